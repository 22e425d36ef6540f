// Device side of the host data path: what util.collate_func_for_dep's torchvision
// transforms (Captioning_models/util.py:13-17,100-101) and DPT_Depthestimator.standardize_depth_map
// (Depth_caption_model/DPT_model.py:43-61) compute, plus the row gather behind the device-resident depth cache
// that replaces the reference's CPU dictionary (Depth_caption_model/depth_train.py:192-202).  All HBM-bound.
#include "dic.h"
#include "common.h"

namespace dic {

// out[b,c,:,:] = (in[b,c,:,:] - mean[c]) / std[c]        (T.Normalize, util.py:13)
__global__ void __launch_bounds__(256) normalize_images_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                long long n, long long hw, int C, float m0, float m1,
                                                                float m2, float s0, float s1, float s2) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const int c = (int)((i / hw) % C);
    const float m = c == 0 ? m0 : (c == 1 ? m1 : m2), s = c == 0 ? s0 : (c == 1 ? s1 : s2);
    out[i] = (in[i] - m) / s;
  }
}

// bilinear resize (align_corners=False, no antialias: identical to torchvision/F.interpolate when up-scaling),
// optional centre crop window, then y = v*mul + add (T.Normalize(0.5,0.5): mul 2, add -1)   (util.py:14-17)
__global__ void __launch_bounds__(256) resize_bilinear_kernel(const float* __restrict__ in, int planes, int H, int W,
                                                               int RH, int RW, int crop_y, int crop_x, int OH, int OW,
                                                               float mul, float add, float* __restrict__ out) {
  const long long total = (long long)planes * OH * OW;
  const long long stride = (long long)gridDim.x * 256;
  const float sy = (float)H / (float)RH, sx = (float)W / (float)RW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int ox = (int)(i % OW);
    const int oy = (int)((i / OW) % OH);
    const long long pl = i / ((long long)OW * OH);
    float fy = ((float)(oy + crop_y) + 0.5f) * sy - 0.5f, fx = ((float)(ox + crop_x) + 0.5f) * sx - 0.5f;
    fy = fmaxf(fy, 0.f); fx = fmaxf(fx, 0.f);
    const int y0 = min((int)fy, H - 1), x0 = min((int)fx, W - 1);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float wy = fy - (float)y0, wx = fx - (float)x0;
    const float* p = in + pl * H * W;
    const float top = p[(long long)y0 * W + x0] * (1.f - wx) + p[(long long)y0 * W + x1] * wx;
    const float bot = p[(long long)y1 * W + x0] * (1.f - wx) + p[(long long)y1 * W + x1] * wx;
    out[i] = (top * (1.f - wy) + bot * wy) * mul + add;
  }
}

// per-image min-max to [0,1] with NaN -> 0.5 first      (DPT_model.py:50-59); one workgroup per image
__global__ void __launch_bounds__(256) depth_standardize_kernel(float* __restrict__ d, long long hw) {
  __shared__ float smin[4], smax[4];
  float* p = d + (long long)blockIdx.x * hw;
  float lo = INFINITY, hi = -INFINITY;
  for (long long i = threadIdx.x; i < hw; i += 256) {
    float v = p[i];
    if (v != v) { v = 0.5f; p[i] = v; }
    lo = fminf(lo, v); hi = fmaxf(hi, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
  if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
  __syncthreads();
  lo = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
  hi = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
  const float dist = hi - lo;
  for (long long i = threadIdx.x; i < hw; i += 256) p[i] = (p[i] - lo) / dist;
}

// out[r,:] = table[idx[r],:]    (rows of `row_floats` floats, row_floats % 4 == 0)
__global__ void __launch_bounds__(256) gather_rows_kernel(const float* __restrict__ table, const long long* __restrict__ idx,
                                                           long long row4, float* __restrict__ out) {
  const float4* src = reinterpret_cast<const float4*>(table) + idx[blockIdx.y] * row4;
  float4* dst = reinterpret_cast<float4*>(out) + (long long)blockIdx.y * row4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < row4; i += (long long)gridDim.x * 256) dst[i] = src[i];
}

__global__ void __launch_bounds__(256) bn_ema_update_kernel(float* __restrict__ running, const float* __restrict__ delta,
                                                             long long n, float keep, const unsigned* __restrict__ skip_if_raised) {
  if (skip_if_raised && *skip_if_raised != 0u) return;       // f16x2 overflow guard (dic.h): a flagged forward's statistics are dropped
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) running[i] = fmaf(keep, running[i], delta[i]);
}

}  // namespace dic

using namespace dic;

extern "C" {

int dic_normalize_images(const float* in, float* out, int B, int C, int H, int W, const float* mean3, const float* std3,
                         void* stream) {
  DIC_REQUIRE(in && out && mean3 && std3 && B > 0 && C >= 1 && C <= 3, "normalize_images: bad arguments");
  const long long hw = (long long)H * W, n = (long long)B * C * hw;
  const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(normalize_images_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, n, hw, C, mean3[0],
                     mean3[C > 1 ? 1 : 0], mean3[C > 2 ? 2 : 0], std3[0], std3[C > 1 ? 1 : 0], std3[C > 2 ? 2 : 0]);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

int dic_resize_bilinear(const float* in, int planes, int H, int W, int resize_short, int crop, float mul, float add,
                        float* out, void* stream) {
  DIC_REQUIRE(in && out && planes > 0 && H > 0 && W > 0 && resize_short > 0 && crop > 0 && crop <= resize_short,
              "resize_bilinear: bad arguments");
  // T.Resize(int): shorter edge -> resize_short keeping the aspect ratio; T.CenterCrop(crop)
  int RH, RW;
  if (H <= W) { RH = resize_short; RW = (int)((long long)resize_short * W / H); }
  else { RW = resize_short; RH = (int)((long long)resize_short * H / W); }
  const int cy = (int)lroundf((RH - crop) / 2.0f), cx = (int)lroundf((RW - crop) / 2.0f);
  const long long total = (long long)planes * crop * crop;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, planes, H, W, RH, RW, cy,
                     cx, crop, crop, mul, add, out);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

int dic_depth_standardize(float* depth, int B, long long hw, void* stream) {
  DIC_REQUIRE(depth && B > 0 && hw > 0, "depth_standardize: bad arguments");
  hipLaunchKernelGGL(depth_standardize_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, depth, hw);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

int dic_gather_rows(const float* table, const int64_t* idx, int n, long long row_floats, float* out, void* stream) {
  DIC_REQUIRE(table && idx && out && n > 0 && row_floats > 0 && row_floats % 4 == 0, "gather_rows: bad arguments");
  const long long row4 = row_floats / 4;
  const int bx = (int)((row4 + 255) / 256 < 64 ? (row4 + 255) / 256 : 64);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(bx, n), dim3(256), 0, (hipStream_t)stream, table, (const long long*)idx, row4,
                     out);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

/* running[i] = (1 - momentum) * running[i] + delta[i]: the BatchNorm running-statistics update of ONE batch applied after the
 * fact.  A frozen-encoder forward that runs ahead on a side stream (engine.prefetch_features) is given zeroed scratch buffers
 * in place of running_mean / running_var, so its finalize kernels leave delta = momentum * batch statistic there; the
 * trainer applies the deltas on the main stream in the order the batches are consumed, which keeps the running statistics
 * in batch order (quirk Q1, depth_train.py:161) however many forwards are in flight. */
int dic_bn_ema_update(float* running, const float* delta, long long n, float momentum, void* stream) {
  return dic_bn_ema_update_guarded(running, delta, n, momentum, nullptr, stream);
}
int dic_bn_ema_update_guarded(float* running, const float* delta, long long n, float momentum, const uint32_t* skip_if_raised,
                              void* stream) {
  DIC_REQUIRE(running && delta && n > 0 && momentum >= 0.f && momentum <= 1.f, "bn_ema_update: bad arguments");
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(bn_ema_update_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, running, delta, n, 1.0f - momentum,
                     (const unsigned*)skip_if_raised);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

}  // extern "C"
