// BatchNorm / pooling / elementwise kernels of the CNN encoders (NHWC, fp32). All HBM-bound:
// 16-byte vector accesses, channel index on the fastest-varying lanes, grid-stride loops.
#include "nn_kernels.h"
#include "gemm.h"      // plane_offset (paired bf16x3 plane layout)
#include <algorithm>

namespace dic {

static inline int ew_blocks(long long n_items) { return (int)std::min<long long>((n_items + 255) / 256, 8192); }

// Division of indices below 2^31 by a runtime constant as one multiply-high: q = (umulhi(x, m) + x) >> s (round-up method; m, s from
// the host).  Until round 4 the pooling kernels spent most of their time in 64-bit divisions (element -> row, channel; row -> image,
// y, x): the 175-MB layer-1 map of the depth encoder moved at 2.5-3.5 TB/s through the BatchNorm-pool backward.
struct FastDiv { unsigned m, s; };
static inline FastDiv make_fastdiv(unsigned d) {
  unsigned s = 0;
  while ((1u << s) < d) ++s;
  return FastDiv{(unsigned)((((unsigned long long)1 << 32) * (((unsigned long long)1 << s) - d)) / d + 1), s};
}
__device__ __forceinline__ unsigned fast_div(unsigned x, FastDiv f) { return (__umulhi(x, f.m) + x) >> f.s; }

// ------------------------------------------------------------------------------------------
// BatchNorm statistics -> scale/shift
// ------------------------------------------------------------------------------------------
// One launch: workgroup = 32 channels x 32 tile-lanes (1024 threads) reduces the conv epilogue's per-M-tile
// partials [mtiles][2][C] in fp64 (fixed order: lane-strided partial sums, then a fixed LDS tree) and produces
// scale/shift (+ saved mean/invstd, running-stat update).  Deep layers have <= 200 M tiles -> <= 7 iterations.
__global__ void __launch_bounds__(1024) bn_finalize_train_kernel(const float* __restrict__ partial, int mtiles,
                                                                  double count, int C, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta,
                                                                  float* __restrict__ rmean, float* __restrict__ rvar,
                                                                  BnBuf out, unsigned* __restrict__ status) {
  __shared__ double s1[32][33], s2[32][33];
  const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double a = 0.0, b = 0.0;
  if (c < C) {
    // the partials were written by other XCDs a moment ago, so every load is a long-latency miss: issue a thread's
    // loads 8 tiles (16 values) at a time instead of chaining them through the fp64 adds
    for (int t = g; t < mtiles; t += 32 * 8) {
      float va[8], vb[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {     // branch-free guard: clamped address, select afterwards (a predicated load
        const int tt = t + u * 32;      // would be followed by vmcnt(0) and serialise the batch)
        const int tc = min(tt, mtiles - 1);
        const float xa = partial[((long long)tc * 2 + 0) * C + c], xb = partial[((long long)tc * 2 + 1) * C + c];
        va[u] = tt < mtiles ? xa : 0.f;
        vb[u] = tt < mtiles ? xb : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { a += (double)va[u]; b += (double)vb[u]; }
    }
  }
  s1[g][cl] = a; s2[g][cl] = b;
  __syncthreads();
  if (g < 4) {     // 32 partials per channel: four groups add 8 each, then group 0 adds the four
    a = 0.0; b = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a += s1[g * 8 + i][cl]; b += s2[g * 8 + i][cl]; }
  }
  __syncthreads();
  if (g < 4) { s1[g][cl] = a; s2[g][cl] = b; }
  __syncthreads();
  if (g == 0 && c < C) {
    a = (s1[0][cl] + s1[1][cl]) + (s1[2][cl] + s1[3][cl]);
    b = (s2[0][cl] + s2[1][cl]) + (s2[2][cl] + s2[3][cl]);
    const double mean = a / count;
    double var = b / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = 1.0f / sqrtf((float)var + kBnEps);
    const float sc = gamma[c] * invstd;
    out.scale[c] = sc;
    out.shift[c] = beta[c] - (float)mean * sc;
    out.mean[c] = (float)mean;
    out.invstd[c] = invstd;
    // an inf / NaN in the convolution output (an overflowed f16x2 plane upstream, a non-finite input image) shows here as non-finite
    // sums: raise the guard word and keep the running statistics of this channel as they were
    const bool finite = fabs(a) <= 1.7e308 && fabs(b) <= 1.7e308;
    if (!finite) f16x2_raise(status, 8u);
    if (rmean && finite) {
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      // running = (1 - m) * running + [m * statistic]: the product in brackets rounded on its own, then ONE fused multiply-add - the form
      // the deferred update takes by construction (delta = m * statistic left in scratch, dic_bn_ema_update adds it), so that a forward
      // run ahead of its batch and one run in place leave the same bits (which of the two products hipcc fuses is otherwise its choice:
      // round 4 found 25 % of the running statistics one ulp apart between the two routes)
      rmean[c] = fmaf(1.f - kBnMomentum, rmean[c], __fmul_rn(kBnMomentum, (float)mean));
      rvar[c] = fmaf(1.f - kBnMomentum, rvar[c], __fmul_rn(kBnMomentum, (float)unb));
    }
  }
}

__global__ void __launch_bounds__(256) bn_finalize_eval_kernel(int C, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta,
                                                                const float* __restrict__ rmean,
                                                                const float* __restrict__ rvar, BnBuf out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.0f / sqrtf(rvar[c] + kBnEps);
  const float sc = gamma[c] * invstd;
  out.scale[c] = sc;
  out.shift[c] = beta[c] - rmean[c] * sc;
  out.mean[c] = rmean[c];
  out.invstd[c] = invstd;
}

// Two-stage variant for layers with thousands of M tiles and few channels (a C/32-block grid would crawl):
// stage 1 reduces slices of 256 tiles to fp64 partials red[S][2][C]; stage 2 is the kernel above on those partials.
constexpr int kBnSliceTiles = 256;
__global__ void __launch_bounds__(256) bn_stats_slice_kernel(const float* __restrict__ partial, int mtiles, int C,
                                                              double* __restrict__ red) {
  __shared__ double s1[8][32], s2[8][32];
  const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int t0 = blockIdx.y * kBnSliceTiles, t1 = min(mtiles, t0 + kBnSliceTiles);
  double a = 0.0, b = 0.0;
  if (c < C)
    for (int t = t0 + g; t < t1; t += 64) {               // 8 tiles (16 values) in flight per thread
      float va[8], vb[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int tt = t + 8 * u;
        const int tc = min(tt, t1 - 1);                    // branch-free guard (see bn_finalize_train_kernel)
        const float xa = partial[((long long)tc * 2 + 0) * C + c], xb = partial[((long long)tc * 2 + 1) * C + c];
        va[u] = tt < t1 ? xa : 0.f;
        vb[u] = tt < t1 ? xb : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { a += (double)va[u]; b += (double)vb[u]; }
    }
  s1[g][cl] = a; s2[g][cl] = b;
  __syncthreads();
  if (g == 0 && c < C) {
#pragma unroll
    for (int i = 1; i < 8; ++i) { a += s1[i][cl]; b += s2[i][cl]; }
    red[((long long)blockIdx.y * 2 + 0) * C + c] = a;
    red[((long long)blockIdx.y * 2 + 1) * C + c] = b;
  }
}

__global__ void __launch_bounds__(256) bn_finalize_from_slices_kernel(const double* __restrict__ red, int slices,
                                                                       double count, int C,
                                                                       const float* __restrict__ gamma,
                                                                       const float* __restrict__ beta,
                                                                       float* __restrict__ rmean, float* __restrict__ rvar,
                                                                       BnBuf out, unsigned* __restrict__ status) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int s = 0; s < slices; ++s) {
    a += red[((long long)s * 2 + 0) * C + c];
    b += red[((long long)s * 2 + 1) * C + c];
  }
  const double mean = a / count;
  double var = b / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = 1.0f / sqrtf((float)var + kBnEps);
  const float sc = gamma[c] * invstd;
  out.scale[c] = sc;
  out.shift[c] = beta[c] - (float)mean * sc;
  out.mean[c] = (float)mean;
  out.invstd[c] = invstd;
  const bool finite = fabs(a) <= 1.7e308 && fabs(b) <= 1.7e308;      // (see bn_finalize_train_kernel)
  if (!finite) f16x2_raise(status, 8u);
  if (rmean && finite) {
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    rmean[c] = fmaf(1.f - kBnMomentum, rmean[c], __fmul_rn(kBnMomentum, (float)mean));      // (see bn_finalize_train_kernel)
    rvar[c] = fmaf(1.f - kBnMomentum, rvar[c], __fmul_rn(kBnMomentum, (float)unb));
  }
}

static int g_bn_two_level_rows = 1024;
void bn_finalize_two_level_rows(int rows) { g_bn_two_level_rows = rows; }
size_t bn_finalize_ws_doubles(long long max_mtiles, int C) {
  return (size_t)ceil_div(max_mtiles, kBnSliceTiles) * 2 * (size_t)C + 16;
}

int bn_finalize_train(const float* partial, int mtiles, long long count, int C, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, BnBuf out, double* red, hipStream_t st, unsigned* status) {
  // (two launches only where one would crawl: ResNet layer 1 has 3136 rows of partials; layer 2's 784 go through the single kernel -
  //  four batches of loads per thread, ~7 us against 5 + 5.4 for the pair; switch g_bn_two_level_rows for the A/B)
  if (mtiles > g_bn_two_level_rows && red != nullptr) {
    const int slices = ceil_div(mtiles, kBnSliceTiles);
    hipLaunchKernelGGL(bn_stats_slice_kernel, dim3(ceil_div(C, 32), slices), dim3(256), 0, st, partial, mtiles, C, red);
    hipLaunchKernelGGL(bn_finalize_from_slices_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, st, (const double*)red,
                       slices, (double)count, C, gamma, beta, running_mean, running_var, out, status);
  } else {
    hipLaunchKernelGGL(bn_finalize_train_kernel, dim3(ceil_div(C, 32)), dim3(1024), 0, st, partial, mtiles,
                       (double)count, C, gamma, beta, running_mean, running_var, out, status);
  }
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

int bn_finalize_eval(int C, const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                     BnBuf out, hipStream_t st) {
  hipLaunchKernelGGL(bn_finalize_eval_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, st, C, gamma, beta, running_mean,
                     running_var, out);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

// ------------------------------------------------------------------------------------------
// y = act(x*scale + shift (+ residual))
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                        float* __restrict__ y, long long n4, int C4, BnBuf bn, int relu) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const int c = (int)(i % C4) * 4;
    float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 s = *reinterpret_cast<const float4*>(bn.scale + c);
    const float4 t = *reinterpret_cast<const float4*>(bn.shift + c);
    v.x = v.x * s.x + t.x; v.y = v.y * s.y + t.y; v.z = v.z * s.z + t.z; v.w = v.w * s.w + t.w;
    if (res) {
      const float4 r = reinterpret_cast<const float4*>(res)[i];
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    reinterpret_cast<float4*>(y)[i] = v;
  }
}

int bn_apply(const float* x, const float* residual, float* y, long long rows, int C, BnBuf bn, int relu,
             hipStream_t st) {
  DIC_REQUIRE(C % 4 == 0, "bn_apply: C %% 4");
  const long long n4 = rows * C / 4;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_blocks(n4)), dim3(256), 0, st, x, residual, y, n4, C / 4, bn, relu);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

// y = act(x*scale + shift (+ residual)) written as three bf16 planes (operand format of the bf16x3 convolution,
// gemm_bf3.hip: row-pair interleaved, plane_offset(..., 1)) and optionally also as fp32 (block outputs are the next
// block's identity).  8 consecutive threads produce one 128-B plane line (4 take the 32 channels of pixel 2q, 4 those of
// pixel 2q+1; 16-B loads and stores throughout - measured 1 % of the ResNet forward against 8-B plane stores).
__global__ void __launch_bounds__(256) bn_apply_planes_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                               const unsigned short* __restrict__ rhi,
                                                               const unsigned short* __restrict__ rmid,
                                                               const unsigned short* __restrict__ rlo,
                                                               float* __restrict__ y, unsigned short* __restrict__ hi,
                                                               unsigned short* __restrict__ mid,
                                                               unsigned short* __restrict__ lo, long long rows, int C,
                                                               BnBuf bn, int relu, const float* __restrict__ rscale,
                                                               const float* __restrict__ rshift, unsigned* __restrict__ status) {
  // one thread = 8 channels of one pixel: 2 x 16-B loads in, one 16-B store per plane out; 8 consecutive threads
  // produce one 128-B plane line (4 threads per pixel of the pair)
  const long long n8 = ((rows + 1) >> 1) * (C / 4);
  const int kb = C / 32;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
    const long long line = i >> 3;
    const int j = (int)(i & 7);
    const long long r = (line / kb) * 2 + (j >> 2);
    const int c = (int)(line % kb) * 32 + (j & 3) * 8;
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = 0.f;
    if (r < rows) {
      const long long src = r * C + c;
      const float4 v0 = *reinterpret_cast<const float4*>(x + src), v1 = *reinterpret_cast<const float4*>(x + src + 4);
      const float4 s0 = *reinterpret_cast<const float4*>(bn.scale + c), s1 = *reinterpret_cast<const float4*>(bn.scale + c + 4);
      const float4 t0 = *reinterpret_cast<const float4*>(bn.shift + c), t1 = *reinterpret_cast<const float4*>(bn.shift + c + 4);
      v[0] = v0.x * s0.x + t0.x; v[1] = v0.y * s0.y + t0.y; v[2] = v0.z * s0.z + t0.z; v[3] = v0.w * s0.w + t0.w;
      v[4] = v1.x * s1.x + t1.x; v[5] = v1.y * s1.y + t1.y; v[6] = v1.z * s1.z + t1.z; v[7] = v1.w * s1.w + t1.w;
      if (res) {
        float4 q0 = *reinterpret_cast<const float4*>(res + src), q1 = *reinterpret_cast<const float4*>(res + src + 4);
        if (rscale) {     // the residual is a raw convolution output with its own BatchNorm (downsample branch)
          const float4 a0 = *reinterpret_cast<const float4*>(rscale + c), a1 = *reinterpret_cast<const float4*>(rscale + c + 4);
          const float4 b0 = *reinterpret_cast<const float4*>(rshift + c), b1 = *reinterpret_cast<const float4*>(rshift + c + 4);
          q0.x = q0.x * a0.x + b0.x; q0.y = q0.y * a0.y + b0.y; q0.z = q0.z * a0.z + b0.z; q0.w = q0.w * a0.w + b0.w;
          q1.x = q1.x * a1.x + b1.x; q1.y = q1.y * a1.y + b1.y; q1.z = q1.z * a1.z + b1.z; q1.w = q1.w * a1.w + b1.w;
        }
        v[0] += q0.x; v[1] += q0.y; v[2] += q0.z; v[3] += q0.w; v[4] += q1.x; v[5] += q1.y; v[6] += q1.z; v[7] += q1.w;
      } else if (rhi) {   // residual given as planes of the same layout: (hi + mid) + lo is the fp32 value, exactly
        const uint4 a = reinterpret_cast<const uint4*>(rhi)[i], b = reinterpret_cast<const uint4*>(rmid)[i],
                    d = reinterpret_cast<const uint4*>(rlo)[i];
        const unsigned aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w}, dw[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[2 * q] += (__uint_as_float(aw[q] << 16) + __uint_as_float(bw[q] << 16)) + __uint_as_float(dw[q] << 16);
          v[2 * q + 1] += (__uint_as_float(aw[q] & 0xffff0000u) + __uint_as_float(bw[q] & 0xffff0000u)) +
                          __uint_as_float(dw[q] & 0xffff0000u);
        }
      }
      if (relu) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
      }
      if (y) {
        *reinterpret_cast<float4*>(y + src) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(y + src + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
    }
    unsigned short h[8], m[8], l[8];
#define DIC_PK(A_, Q_) ((unsigned)A_[Q_] | ((unsigned)A_[Q_ + 1] << 16))
    if (!lo) {      // f16x2 format (gemm_bf3.hip): two fp16 planes of kF16ActScale * v
      bool bad = false;
#pragma unroll
      for (int q = 0; q < 8; ++q) bad |= f16x2_out_of_range(v[q], kF16ActScale);
      if (bad) f16x2_raise(status);
#pragma unroll
      for (int q = 0; q < 8; ++q) split2_f16(v[q], kF16ActScale, h[q], m[q]);
      reinterpret_cast<uint4*>(hi)[i] = make_uint4(DIC_PK(h, 0), DIC_PK(h, 2), DIC_PK(h, 4), DIC_PK(h, 6));
      reinterpret_cast<uint4*>(mid)[i] = make_uint4(DIC_PK(m, 0), DIC_PK(m, 2), DIC_PK(m, 4), DIC_PK(m, 6));
      continue;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) split3_bf16(v[q], h[q], m[q], l[q]);
    reinterpret_cast<uint4*>(hi)[i] = make_uint4(DIC_PK(h, 0), DIC_PK(h, 2), DIC_PK(h, 4), DIC_PK(h, 6));
    reinterpret_cast<uint4*>(mid)[i] = make_uint4(DIC_PK(m, 0), DIC_PK(m, 2), DIC_PK(m, 4), DIC_PK(m, 6));
    reinterpret_cast<uint4*>(lo)[i] = make_uint4(DIC_PK(l, 0), DIC_PK(l, 2), DIC_PK(l, 4), DIC_PK(l, 6));
#undef DIC_PK
  }
}

int bn_apply_planes(const float* x, const float* residual, const unsigned short* const residual_planes[3], float* y,
                    unsigned short* const planes[3], long long rows, int C, BnBuf bn, int relu, hipStream_t st,
                    const BnBuf* residual_bn, unsigned* status) {
  DIC_REQUIRE(!residual_bn || residual, "bn_apply_planes: residual_bn needs an fp32 residual");
  DIC_REQUIRE(!(residual && residual_planes), "bn_apply_planes: give the residual as fp32 or as planes, not both");
  DIC_REQUIRE(C % 32 == 0, "bn_apply_planes: C %% 32");
  DIC_REQUIRE(planes[2] || !residual_planes, "bn_apply_planes: f16x2 planes (planes[2] == NULL) are not exact - the residual must be fp32");
  const long long n4 = ((rows + 1) >> 1) * (C / 4);      // threads (8 channels each)
  const unsigned short* r0 = residual_planes ? residual_planes[0] : nullptr;
  const unsigned short* r1 = residual_planes ? residual_planes[1] : nullptr;
  const unsigned short* r2 = residual_planes ? residual_planes[2] : nullptr;
  hipLaunchKernelGGL(bn_apply_planes_kernel, dim3(ew_blocks(n4)), dim3(256), 0, st, x, residual, r0, r1, r2, y, planes[0],
                     planes[1], planes[2], rows, C, bn, relu, residual_bn ? residual_bn->scale : nullptr,
                     residual_bn ? residual_bn->shift : nullptr, status);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

// ------------------------------------------------------------------------------------------
// fused BN + ReLU + max pooling (records the argmax position for the backward pass)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bn_relu_maxpool_kernel(const float* __restrict__ x, int B, int H, int W, int C4,
                                                               BnBuf bn, int has_bn, int relu, int k, int s, int p,
                                                               int PH, int PW, float* __restrict__ y,
                                                               unsigned char* __restrict__ idx,
                                                               unsigned short* __restrict__ hi,
                                                               unsigned short* __restrict__ mid,
                                                               unsigned short* __restrict__ lo, unsigned* __restrict__ status,
                                                               FastDiv dC4, FastDiv dPW, FastDiv dPH, float* __restrict__ xsel) {      // xsel (nullable): the RAW input value at each window's argmax
  const unsigned total = (unsigned)B * PH * PW * C4;       // < 2^31 (checked by the host)
  const unsigned stride = gridDim.x * 256u;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += stride) {
    const unsigned row = fast_div(i, dC4);
    const int c4 = (int)(i - row * (unsigned)C4);
    const unsigned r1 = fast_div(row, dPW);
    const int pw = (int)(row - r1 * (unsigned)PW);
    const int b = (int)fast_div(r1, dPH);
    const int ph = (int)(r1 - (unsigned)b * (unsigned)PH);
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (has_bn) {
      sc = *reinterpret_cast<const float4*>(bn.scale + c4 * 4);
      sh = *reinterpret_cast<const float4*>(bn.shift + c4 * 4);
    }
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    float4 braw = make_float4(0.f, 0.f, 0.f, 0.f);
    uchar4 bi = make_uchar4(0, 0, 0, 0);
    for (int kh = 0; kh < k; ++kh) {
      const int h = ph * s - p + kh;
      if ((unsigned)h >= (unsigned)H) continue;
      for (int kw = 0; kw < k; ++kw) {
        const int w = pw * s - p + kw;
        if ((unsigned)w >= (unsigned)W) continue;
        float4 v = reinterpret_cast<const float4*>(x)[(((long long)b * H + h) * W + w) * C4 + c4];
        const float4 raw = v;
        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        const unsigned char id = (unsigned char)(kh * k + kw);
        if (v.x > best.x) { best.x = v.x; bi.x = id; braw.x = raw.x; }
        if (v.y > best.y) { best.y = v.y; bi.y = id; braw.y = raw.y; }
        if (v.z > best.z) { best.z = v.z; bi.z = id; braw.z = raw.z; }
        if (v.w > best.w) { best.w = v.w; bi.w = id; braw.w = raw.w; }
      }
    }
    if (y) reinterpret_cast<float4*>(y)[i] = best;
    if (idx) reinterpret_cast<uchar4*>(idx)[i] = bi;
    if (xsel) reinterpret_cast<float4*>(xsel)[i] = braw;
    if (hi) {     // also (or only) as paired bf16x3 planes: the pooled map feeds a bf16x3 convolution (saves the split pass)
      const long long off = plane_offset((long long)row, c4 * 4, C4 / 8, 1);
      unsigned short h[4], m[4], l[4];
      if (!lo) {    // f16x2 format: two fp16 planes of kF16ActScale * v
        if (f16x2_out_of_range(best.x, kF16ActScale) | f16x2_out_of_range(best.y, kF16ActScale) | f16x2_out_of_range(best.z, kF16ActScale) |
            f16x2_out_of_range(best.w, kF16ActScale)) f16x2_raise(status, 2u);
        split2_f16(best.x, kF16ActScale, h[0], m[0]); split2_f16(best.y, kF16ActScale, h[1], m[1]);
        split2_f16(best.z, kF16ActScale, h[2], m[2]); split2_f16(best.w, kF16ActScale, h[3], m[3]);
      } else {
        split3_bf16(best.x, h[0], m[0], l[0]); split3_bf16(best.y, h[1], m[1], l[1]);
        split3_bf16(best.z, h[2], m[2], l[2]); split3_bf16(best.w, h[3], m[3], l[3]);
      }
      *reinterpret_cast<uint2*>(hi + off) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
      *reinterpret_cast<uint2*>(mid + off) = make_uint2((unsigned)m[0] | ((unsigned)m[1] << 16), (unsigned)m[2] | ((unsigned)m[3] << 16));
      if (lo) *reinterpret_cast<uint2*>(lo + off) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
    }
  }
}

int bn_relu_maxpool(const float* x, int B, int H, int W, int C, const BnBuf* bn, int relu, int k, int s, int p,
                    float* y, unsigned char* idx, hipStream_t st, unsigned short* const planes[3], unsigned* status, float* xsel) {
  DIC_REQUIRE(C % 4 == 0, "maxpool: C %% 4");
  DIC_REQUIRE(y || planes, "maxpool: no output");
  DIC_REQUIRE(!planes || C % 32 == 0, "maxpool: plane output needs C %% 32");
  const int PH = (H + 2 * p - k) / s + 1, PW = (W + 2 * p - k) / s + 1;
  const long long total = (long long)B * PH * PW * (C / 4);
  DIC_REQUIRE(total < (1ll << 31) - (1ll << 22), "maxpool: more than 2^31 output elements");      // (32-bit indices: i + grid stride must not wrap)
  BnBuf z{};
  hipLaunchKernelGGL(bn_relu_maxpool_kernel, dim3(ew_blocks(total)), dim3(256), 0, st, x, B, H, W, C / 4, bn ? *bn : z,
                     bn ? 1 : 0, relu, k, s, p, PH, PW, y, idx, planes ? planes[0] : nullptr, planes ? planes[1] : nullptr,
                     planes ? planes[2] : nullptr, status, make_fastdiv((unsigned)(C / 4)), make_fastdiv((unsigned)PW), make_fastdiv((unsigned)PH), xsel);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

// the loud end of the f16x2 overflow guard (common.h): every element of y becomes NaN when the status word is raised
__global__ void __launch_bounds__(256) poison_if_raised_kernel(float* __restrict__ y, long long n, const unsigned* __restrict__ status) {
  if (*status == 0u) return;
  const float nan = __uint_as_float(0x7fc00000u);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = nan;
}
// Clears the guard word.  A kernel, not hipMemsetAsync: the forward is captured into hipGraphs that several streams (and, with ranks
// sharing a GPU, several processes) replay concurrently, and a captured 256-byte memset node was observed to leave bit patterns of
// other data in the word on replay (gpurun_out/r04_dbg.log: slot words 0x421ffc00 after the second and third slot's replays, zero with
// DIC_RESNET_GRAPH=0) - kernel nodes replay exactly what was captured.
__global__ void clear_status_kernel(unsigned* __restrict__ status) { status[threadIdx.x] = 0u; }
int clear_status(unsigned* status, hipStream_t st) {
  hipLaunchKernelGGL(clear_status_kernel, dim3(1), dim3(64), 0, st, status);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}
int poison_if_raised(float* y, long long n, const unsigned* status, hipStream_t st) {
  hipLaunchKernelGGL(poison_if_raised_kernel, dim3(64), dim3(256), 0, st, y, n, status);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

// ------------------------------------------------------------------------------------------
// adaptive average pooling to OUT x OUT (window [floor(i*H/OUT), ceil((i+1)*H/OUT)) )
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) adaptive_avgpool_kernel(const float* __restrict__ x, int B, int H, int W, int C4,
                                                                BnBuf bn, int has_bn, int relu, int OUT,
                                                                float* __restrict__ y) {
  const long long total = (long long)B * OUT * OUT * C4;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int c4 = (int)(i % C4);
    long long r = i / C4;
    const int oj = (int)(r % OUT); r /= OUT;
    const int oi = (int)(r % OUT);
    const int b = (int)(r / OUT);
    const int h0 = (oi * H) / OUT, h1 = ((oi + 1) * H + OUT - 1) / OUT;
    const int w0 = (oj * W) / OUT, w1 = ((oj + 1) * W + OUT - 1) / OUT;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (has_bn) {
      sc = *reinterpret_cast<const float4*>(bn.scale + c4 * 4);
      sh = *reinterpret_cast<const float4*>(bn.shift + c4 * 4);
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int h = h0; h < h1; ++h)
      for (int w = w0; w < w1; ++w) {
        float4 v = reinterpret_cast<const float4*>(x)[(((long long)b * H + h) * W + w) * C4 + c4];
        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    const float n = (float)((h1 - h0) * (w1 - w0));
    acc.x /= n; acc.y /= n; acc.z /= n; acc.w /= n;
    reinterpret_cast<float4*>(y)[i] = acc;
  }
}

int adaptive_avgpool(const float* x, int B, int H, int W, int C, const BnBuf* bn, int relu, int out, float* y,
                     hipStream_t st) {
  DIC_REQUIRE(C % 4 == 0, "avgpool: C %% 4");
  const long long total = (long long)B * out * out * (C / 4);
  BnBuf z{};
  hipLaunchKernelGGL(adaptive_avgpool_kernel, dim3(ew_blocks(total)), dim3(256), 0, st, x, B, H, W, C / 4,
                     bn ? *bn : z, bn ? 1 : 0, relu, out, y);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

__global__ void __launch_bounds__(256) adaptive_avgpool_bwd_kernel(const float* __restrict__ dy, int B, int H, int W,
                                                                    int C4, int OUT, float* __restrict__ dx) {
  const long long total = (long long)B * H * W * C4;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int c4 = (int)(i % C4);
    long long r = i / C4;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int b = (int)(r / H);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int oi_lo = max(0, (h * OUT) / H - 1), oi_hi = min(OUT - 1, ((h + 1) * OUT + H - 1) / H + 1);
    const int oj_lo = max(0, (w * OUT) / W - 1), oj_hi = min(OUT - 1, ((w + 1) * OUT + W - 1) / W + 1);
    for (int oi = oi_lo; oi <= oi_hi; ++oi) {
      const int h0 = (oi * H) / OUT, h1 = ((oi + 1) * H + OUT - 1) / OUT;
      if (h < h0 || h >= h1) continue;
      for (int oj = oj_lo; oj <= oj_hi; ++oj) {
        const int w0 = (oj * W) / OUT, w1 = ((oj + 1) * W + OUT - 1) / OUT;
        if (w < w0 || w >= w1) continue;
        const float inv = 1.0f / (float)((h1 - h0) * (w1 - w0));
        const float4 g = reinterpret_cast<const float4*>(dy)[(((long long)b * OUT + oi) * OUT + oj) * C4 + c4];
        acc.x += g.x * inv; acc.y += g.y * inv; acc.z += g.z * inv; acc.w += g.w * inv;
      }
    }
    reinterpret_cast<float4*>(dx)[i] = acc;
  }
}

int adaptive_avgpool_bwd(const float* dy, int B, int H, int W, int C, int out, float* dx, hipStream_t st) {
  const long long total = (long long)B * H * W * (C / 4);
  hipLaunchKernelGGL(adaptive_avgpool_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, st, dy, B, H, W, C / 4, out, dx);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

// ------------------------------------------------------------------------------------------
// backward helpers
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) maxpool_relu_bwd_kernel(const float* __restrict__ dpool,
                                                                const unsigned char* __restrict__ idx,
                                                                const float* __restrict__ x, int B, int H, int W, int C4,
                                                                int k, int PH, int PW, BnBuf bn, float* __restrict__ dy) {
  const long long total = (long long)B * H * W * C4;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int c4 = (int)(i % C4);
    long long r = i / C4;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int b = (int)(r / H);
    const int ph = h / k, pw = w / k;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ph < PH && pw < PW) {
      const long long po = (((long long)b * PH + ph) * PW + pw) * C4 + c4;
      const float4 d = reinterpret_cast<const float4*>(dpool)[po];
      const uchar4 id = reinterpret_cast<const uchar4*>(idx)[po];
      const unsigned char me = (unsigned char)((h - ph * k) * k + (w - pw * k));
      const float4 v = reinterpret_cast<const float4*>(x)[i];
      const float4 sc = *reinterpret_cast<const float4*>(bn.scale + c4 * 4);
      const float4 sh = *reinterpret_cast<const float4*>(bn.shift + c4 * 4);
      g.x = (id.x == me && v.x * sc.x + sh.x > 0.f) ? d.x : 0.f;
      g.y = (id.y == me && v.y * sc.y + sh.y > 0.f) ? d.y : 0.f;
      g.z = (id.z == me && v.z * sc.z + sh.z > 0.f) ? d.z : 0.f;
      g.w = (id.w == me && v.w * sc.w + sh.w > 0.f) ? d.w : 0.f;
    }
    reinterpret_cast<float4*>(dy)[i] = g;
  }
}

int maxpool_relu_bwd(const float* dpool, const unsigned char* idx, const float* x, int B, int H, int W, int C, int k,
                     BnBuf bn, float* dy, hipStream_t st) {
  const long long total = (long long)B * H * W * (C / 4);
  hipLaunchKernelGGL(maxpool_relu_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, st, dpool, idx, x, B, H, W, C / 4,
                     k, H / k, W / k, bn, dy);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

// BN + ReLU decision of one element: the fused multiply-add the forward kernels contract `v * scale + shift` into
__device__ __forceinline__ bool relu_passes(float v, float sc, float sh) { return fmaf(v, sc, sh) > 0.f; }

__global__ void __launch_bounds__(256) relu_mask_bwd_kernel(float* __restrict__ dy, const float* __restrict__ x,
                                                             long long n4, int C4, BnBuf bn) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const int c = (int)(i % C4) * 4;
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 sc = *reinterpret_cast<const float4*>(bn.scale + c);
    const float4 sh = *reinterpret_cast<const float4*>(bn.shift + c);
    float4 g = reinterpret_cast<float4*>(dy)[i];
    if (!relu_passes(v.x, sc.x, sh.x)) g.x = 0.f;
    if (!relu_passes(v.y, sc.y, sh.y)) g.y = 0.f;
    if (!relu_passes(v.z, sc.z, sh.z)) g.z = 0.f;
    if (!relu_passes(v.w, sc.w, sh.w)) g.w = 0.f;
    reinterpret_cast<float4*>(dy)[i] = g;
  }
}

// diagnostic (dic_depth_encoder_inspect): the ReLU decisions exactly as relu_mask_bwd_kernel takes them
__global__ void __launch_bounds__(256) relu_mask_export_kernel(const float* __restrict__ x, long long n, int C, BnBuf bn,
                                                                unsigned char* __restrict__ out) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const int c = (int)(i % C);
    out[i] = relu_passes(x[i], bn.scale[c], bn.shift[c]) ? 1 : 0;
  }
}

int relu_mask_export(const float* x, long long rows, int C, BnBuf bn, unsigned char* out, hipStream_t st) {
  hipLaunchKernelGGL(relu_mask_export_kernel, dim3(ew_blocks(rows * C)), dim3(256), 0, st, x, rows * C, C, bn, out);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

int relu_mask_bwd(float* dy, const float* x, long long rows, int C, BnBuf bn, hipStream_t st) {
  const long long n4 = rows * C / 4;
  hipLaunchKernelGGL(relu_mask_bwd_kernel, dim3(ew_blocks(n4)), dim3(256), 0, st, dy, x, n4, C / 4, bn);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

// ------------------------------------------------------------------------------------------
// device-resident f16x2 scales (F16Scale, nn_kernels.h)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) f16_scale_reset_kernel(unsigned* __restrict__ bounds, int n) {
  if ((int)threadIdx.x < n) bounds[threadIdx.x] = 0u;
}
__global__ void __launch_bounds__(256) absmax_kernel(const float* __restrict__ x, long long n4, unsigned* __restrict__ bound) {
  __shared__ float sm[4];
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    if (v.x != v.x || v.y != v.y || v.z != v.z || v.w != v.w) m = __uint_as_float(0x7f800000u);      // a NaN must not hide behind fmaxf
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(bound, __float_as_uint(fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]))));      // one atomic per block
}
// s = 2^(13 - e), e = floor(log2 bound) clamped to [-60, 60] (a zero tensor gets 2^73: its planes are zeros whatever the scale)
__global__ void __launch_bounds__(64) f16_scale_finish_kernel(const unsigned* __restrict__ bound, float* __restrict__ slot) {
  if (threadIdx.x != 0) return;
  const int e = min(60, max(-60, (int)((*bound >> 23) & 0xffu) - 127));
  slot[0] = __uint_as_float((unsigned)(13 - e + 127) << 23);
  slot[1] = __uint_as_float((unsigned)(e - 13 + 127) << 23);
}
int f16_scale_reset(unsigned* bounds, int n, hipStream_t st) {
  DIC_REQUIRE(n <= 64, "f16_scale_reset: at most 64 words");
  hipLaunchKernelGGL(f16_scale_reset_kernel, dim3(1), dim3(64), 0, st, bounds, n);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}
int f16_scale_finish(F16Scale sl, hipStream_t st) {
  hipLaunchKernelGGL(f16_scale_finish_kernel, dim3(1), dim3(64), 0, st, (const unsigned*)sl.bound, sl.slot);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}
int f16_scale_from_absmax(const float* x, long long n, F16Scale sl, hipStream_t st) {
  DIC_REQUIRE(n % 4 == 0, "f16_scale_from_absmax: n %% 4");
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)std::min<long long>((n / 4 + 255) / 256, 128)), dim3(256), 0, st, x, n / 4, sl.bound);
  DIC_LAUNCH_CHECK();
  return f16_scale_finish(sl, st);
}

// BatchNorm backward, stage 1: per (row-chunk, channel) sums of dy and dy*xhat.
// block = 64 channels (16 float4 lanes) x 16 row lanes; grid (C/64, chunks) with enough row chunks for >= 512 blocks
constexpr int kBnChunksMax = 512;      // (256 until round 4: 512 workgroups of 4 rows in flight moved the depth encoder's 175-MB layer-1 map at 2.5 TB/s)
static inline int reduce_chunks(int C) { return std::min(kBnChunksMax, std::max(64, 1024 / std::max(1, C / 64))); }
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             long long rows, int C, BnBuf bn, float* __restrict__ part,
                                                             float* __restrict__ gmaxp) {
  __shared__ float4 sa[16][16], sb[16][16];
  __shared__ float smax[4];
  float gm = 0.f;                        // max |dy| over this block's elements (f16x2 scale bound, nullable)
  const int c4l = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + c4l * 4;
  const long long per = (rows + gridDim.y - 1) / gridDim.y;
  const long long r0 = (long long)blockIdx.y * per, r1 = min(rows, r0 + per);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  if (c < C) {
    const float4 mu = *reinterpret_cast<const float4*>(bn.mean + c);
    const float4 is = *reinterpret_cast<const float4*>(bn.invstd + c);
    for (long long r = r0 + rl; r < r1; r += 64) {        // 4 rows x 2 tensors in flight per thread
      float4 g[4], v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long rr = r + 16 * u;
        const long long rc = min(rr, r1 - 1);              // branch-free guard: clamped row, zeroed afterwards
        g[u] = *reinterpret_cast<const float4*>(dy + rc * C + c);
        v[u] = *reinterpret_cast<const float4*>(x + rc * C + c);
        if (rr >= r1) g[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a.x += g[u].x; a.y += g[u].y; a.z += g[u].z; a.w += g[u].w;
        b.x += g[u].x * (v[u].x - mu.x) * is.x; b.y += g[u].y * (v[u].y - mu.y) * is.y;
        b.z += g[u].z * (v[u].z - mu.z) * is.z; b.w += g[u].w * (v[u].w - mu.w) * is.w;
        gm = fmaxf(fmaxf(gm, fmaxf(fabsf(g[u].x), fabsf(g[u].y))), fmaxf(fabsf(g[u].z), fabsf(g[u].w)));
      }
    }
  }
  sa[rl][c4l] = a; sb[rl][c4l] = b;
  if (gmaxp) { gm = wave_max(gm); if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = gm; }
  __syncthreads();
  if (gmaxp && threadIdx.x == 0) gmaxp[(long long)blockIdx.y * gridDim.x + blockIdx.x] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
  if (rl == 0 && c < C) {
#pragma unroll
    for (int i = 1; i < 16; ++i) {
      a.x += sa[i][c4l].x; a.y += sa[i][c4l].y; a.z += sa[i][c4l].z; a.w += sa[i][c4l].w;
      b.x += sb[i][c4l].x; b.y += sb[i][c4l].y; b.z += sb[i][c4l].z; b.w += sb[i][c4l].w;
    }
    *reinterpret_cast<float4*>(part + ((long long)blockIdx.y * 2 + 0) * C + c) = a;
    *reinterpret_cast<float4*>(part + ((long long)blockIdx.y * 2 + 1) * C + c) = b;
  }
}

// block = 32 channels x 8 chunk lanes: a channel's `chunks` partial pairs are read 8 at a time by its 8 lanes (all loads of a lane in
// flight together), summed in fp64 per lane and combined in a fixed order - 5 us instead of the 20 us of one thread walking 256 chunks
// (round 4: this kernel runs three times per step on the main stream, whose time adds to the step one for one).
__global__ void __launch_bounds__(256) bn_bwd_finalize_kernel(const float* __restrict__ part, int chunks, int C,
                                                               double rows, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, float* __restrict__ k2,
                                                               float* __restrict__ k3, const float* __restrict__ gmaxp,
                                                               const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                               unsigned* __restrict__ bound) {
  __shared__ double sa[8][32], sb[8][32];
  __shared__ float sg[8][32];
  const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double a = 0.0, b = 0.0;
  float gm = 0.f;
  if (c < C) {
    const int cb = c >> 6, ncb = C >> 6;
    for (int t0 = g; t0 < chunks; t0 += 64) {            // 8 chunks (16 values) in flight per thread
      float va[8], vb[8], vg[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + 8 * u, tc = min(t, chunks - 1);     // branch-free guard
        va[u] = part[((long long)tc * 2 + 0) * C + c];
        vb[u] = part[((long long)tc * 2 + 1) * C + c];
        vg[u] = bound ? gmaxp[(long long)tc * ncb + cb] : 0.f;
        if (t >= chunks) { va[u] = 0.f; vb[u] = 0.f; vg[u] = 0.f; }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { a += (double)va[u]; b += (double)vb[u]; gm = fmaxf(gm, vg[u]); }
    }
  }
  sa[g][cl] = a; sb[g][cl] = b; sg[g][cl] = gm;
  __syncthreads();
  float bnd = 0.f;
  if (g == 0 && c < C) {
#pragma unroll
    for (int i = 1; i < 8; ++i) { a += sa[i][cl]; b += sb[i][cl]; gm = fmaxf(gm, sg[i][cl]); }
    dbeta[c] = (float)a;
    dgamma[c] = (float)b;
    k2[c] = (float)(a / rows);
    k3[c] = (float)(b / rows);
    if (bound) {
      // f16x2 scale of the gradient this backward produces (the apply kernel splits it in the same pass that forms it, so its
      // magnitude has to be bounded beforehand):  dx = gamma * invstd * (g - k2 - xhat * k3),  |xhat| <= sqrt(rows - 1)
      //   =>  |dx| <= |gamma * invstd| * (max |g| + |k2| + sqrt(rows) * |k3|),   max |g| over this channel's 64-channel block.
      // Loose by the xhat bound only (a few powers of two), which the fp16 exponent range absorbs (F16Scale, nn_kernels.h).
      bnd = fabsf(gamma[c] * invstd[c]) * (gm + fabsf((float)(a / rows)) + sqrtf((float)rows) * fabsf((float)(b / rows)));
      if (bnd != bnd) bnd = __uint_as_float(0x7f800000u);
    }
  }
  if (bound && threadIdx.x < 64) {                         // (all of wave 0: the lanes of g == 1 hold zeros)
    bnd = wave_max(bnd);
    if (threadIdx.x == 0) atomicMax(bound, __float_as_uint(bnd));
  }
}

__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(float* __restrict__ dy, const float* __restrict__ x,
                                                            long long n4, int C4, const float* __restrict__ gamma,
                                                            BnBuf bn, const float* __restrict__ k2,
                                                            const float* __restrict__ k3,
                                                            unsigned short* __restrict__ hi,
                                                            unsigned short* __restrict__ mid,
                                                            unsigned short* __restrict__ lo, const float* __restrict__ f16_slot) {
  const long long stride = (long long)gridDim.x * 256;
  const float fs = f16_slot ? f16_slot[0] : 1.f;      // f16x2 planes (lo == NULL): the device-resident scale of this gradient
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const int c = (int)(i % C4) * 4;
    const float4 g = reinterpret_cast<float4*>(dy)[i];
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 mu = *reinterpret_cast<const float4*>(bn.mean + c);
    const float4 is = *reinterpret_cast<const float4*>(bn.invstd + c);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
    const float4 a2 = *reinterpret_cast<const float4*>(k2 + c);
    const float4 a3 = *reinterpret_cast<const float4*>(k3 + c);
    float4 o;
    o.x = ga.x * is.x * (g.x - a2.x - (v.x - mu.x) * is.x * a3.x);
    o.y = ga.y * is.y * (g.y - a2.y - (v.y - mu.y) * is.y * a3.y);
    o.z = ga.z * is.z * (g.z - a2.z - (v.z - mu.z) * is.z * a3.z);
    o.w = ga.w * is.w * (g.w - a2.w - (v.w - mu.w) * is.w * a3.w);
    reinterpret_cast<float4*>(dy)[i] = o;
    if (hi) {     // the same gradient as paired bf16x3 planes (operand of the data-gradient convolution)
      const long long off = plane_offset(i / C4, c, C4 / 8, 1);
      unsigned short h[4], m[4], l[4];
      if (!lo) {      // f16x2 format: two fp16 planes of fs * o
        split2_f16(o.x, fs, h[0], m[0]); split2_f16(o.y, fs, h[1], m[1]); split2_f16(o.z, fs, h[2], m[2]); split2_f16(o.w, fs, h[3], m[3]);
      } else {
        split3_bf16(o.x, h[0], m[0], l[0]); split3_bf16(o.y, h[1], m[1], l[1]);
        split3_bf16(o.z, h[2], m[2], l[2]); split3_bf16(o.w, h[3], m[3], l[3]);
      }
      *reinterpret_cast<uint2*>(hi + off) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
      *reinterpret_cast<uint2*>(mid + off) = make_uint2((unsigned)m[0] | ((unsigned)m[1] << 16), (unsigned)m[2] | ((unsigned)m[3] << 16));
      if (lo) *reinterpret_cast<uint2*>(lo + off) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
    }
  }
}

size_t bn_backward_ws_floats(int C) { return (size_t)kBnChunksMax * 2 * C + 2 * (size_t)C + (size_t)kBnChunksMax * (C / 64 + 1); }

int bn_backward(float* dy_dx, const float* x, long long rows, int C, const float* gamma, BnBuf bn, float* dgamma,
                float* dbeta, float* ws, hipStream_t st, unsigned short* const dx_planes[3], F16Scale* f16) {
  DIC_REQUIRE(C % 64 == 0, "bn_backward: C %% 64");
  const bool f16x2 = dx_planes && !dx_planes[2];
  DIC_REQUIRE(!f16x2 || (f16 && f16->bound && f16->slot), "bn_backward: f16x2 planes need a scale slot");
  float* part = ws;
  float* k2 = ws + (size_t)kBnChunksMax * 2 * C;
  float* k3 = k2 + C;
  float* gmaxp = k3 + C;
  const int chunks = reduce_chunks(C);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C / 64, chunks), dim3(256), 0, st, dy_dx, x, rows, C, bn, part, f16x2 ? gmaxp : nullptr);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, 32)), dim3(256), 0, st, part, chunks, C, (double)rows,
                     dgamma, dbeta, k2, k3, (const float*)gmaxp, gamma, (const float*)bn.invstd, f16x2 ? f16->bound : nullptr);
  if (f16x2) DIC_TRY(f16_scale_finish(*f16, st));
  const long long n4 = rows * C / 4;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_blocks(n4)), dim3(256), 0, st, dy_dx, x, n4, C / 4, gamma, bn, k2, k3,
                     dx_planes ? dx_planes[0] : nullptr, dx_planes ? dx_planes[1] : nullptr,
                     dx_planes ? dx_planes[2] : nullptr, f16x2 ? (const float*)f16->slot : nullptr);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

// pieces of bn_backward for callers that produce the partial sums themselves (depth_layer1.hip): the workspace layout and the finalize
void bn_backward_ws_layout(float* ws, int C, float** part, float** k2, float** k3) {
  *part = ws; *k2 = ws + (size_t)kBnChunksMax * 2 * C; *k3 = *k2 + C;
}
int bn_backward_finalize(const float* part, int chunks, int C, double rows, float* dgamma, float* dbeta, float* k2, float* k3, hipStream_t st) {
  DIC_REQUIRE(chunks >= 1 && chunks <= kBnChunksMax && C % 64 == 0, "bn_backward_finalize: chunks <= %d, C %% 64", kBnChunksMax);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, 32)), dim3(256), 0, st, part, chunks, C, rows, dgamma, dbeta, k2, k3,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (unsigned*)nullptr);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

// ------------------------------------------------------------------------------------------
// Max-pool + ReLU + BatchNorm backward without materialising the (8/9 zero) pooled gradient: the gradient that
// reaches position (b,h,w,c) of the pre-pool tensor is dpool[window] if this position was the window's argmax and
// the BN output was positive, else 0 (non-overlapping k x k windows) - recomputed on the fly in both BN-backward
// passes.  Versus maxpool_relu_bwd + bn_backward this drops one write and two reads of the full-size tensor
// (layer 1: 175 MB each).
// ------------------------------------------------------------------------------------------
struct PoolGeom { int H, W, k, PH, PW; FastDiv dH, dW, dk; };

__device__ __forceinline__ float4 pool_relu_grad4(const float* __restrict__ dpool, const unsigned char* __restrict__ idx,
                                                  const float4 v, const float4 sc, const float4 sh, unsigned row,
                                                  int c4, int C4, PoolGeom pg) {      // row < 2^31 (checked by the host)
  const unsigned r2 = fast_div(row, pg.dW);
  const int w = (int)(row - r2 * (unsigned)pg.W);
  const unsigned b = fast_div(r2, pg.dH);
  const int h = (int)(r2 - b * (unsigned)pg.H);
  const int ph = (int)fast_div((unsigned)h, pg.dk), pw = (int)fast_div((unsigned)w, pg.dk);
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ph < pg.PH && pw < pg.PW) {
    const long long po = (((long long)b * pg.PH + ph) * pg.PW + pw) * C4 + c4;
    const float4 d = reinterpret_cast<const float4*>(dpool)[po];
    const uchar4 id = reinterpret_cast<const uchar4*>(idx)[po];
    const unsigned char me = (unsigned char)((h - ph * pg.k) * pg.k + (w - pw * pg.k));
    g.x = (id.x == me && v.x * sc.x + sh.x > 0.f) ? d.x : 0.f;
    g.y = (id.y == me && v.y * sc.y + sh.y > 0.f) ? d.y : 0.f;
    g.z = (id.z == me && v.z * sc.z + sh.z > 0.f) ? d.z : 0.f;
    g.w = (id.w == me && v.w * sc.w + sh.w > 0.f) ? d.w : 0.f;
  }
  return g;
}

__global__ void __launch_bounds__(256) bn_pool_bwd_reduce_kernel(const float* __restrict__ dpool,
                                                                  const unsigned char* __restrict__ idx,
                                                                  const float* __restrict__ x, long long rows, int C,
                                                                  PoolGeom pg, BnBuf bn, float* __restrict__ part,
                                                                  float* __restrict__ gmaxp) {
  __shared__ float4 sa[16][16], sb[16][16];
  __shared__ float smax[4];
  float gm = 0.f;                        // max |g| over this block's elements (f16x2 scale bound, nullable)
  const int c4l = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + c4l * 4;
  const long long per = (rows + gridDim.y - 1) / gridDim.y;
  const long long r0 = (long long)blockIdx.y * per, r1 = min(rows, r0 + per);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  if (c < C) {
    const float4 mu = *reinterpret_cast<const float4*>(bn.mean + c);
    const float4 is = *reinterpret_cast<const float4*>(bn.invstd + c);
    const float4 sc = *reinterpret_cast<const float4*>(bn.scale + c);
    const float4 sh = *reinterpret_cast<const float4*>(bn.shift + c);
    for (long long r = r0 + rl; r < r1; r += 128) {       // 8 rows in flight per thread
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(x + min(r + 16 * u, r1 - 1) * C + c);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long long rr = r + 16 * u;
        if (rr < r1) {
          const float4 g = pool_relu_grad4(dpool, idx, v[u], sc, sh, (unsigned)rr, c >> 2, C >> 2, pg);
          a.x += g.x; a.y += g.y; a.z += g.z; a.w += g.w;
          b.x += g.x * (v[u].x - mu.x) * is.x; b.y += g.y * (v[u].y - mu.y) * is.y;
          b.z += g.z * (v[u].z - mu.z) * is.z; b.w += g.w * (v[u].w - mu.w) * is.w;
          gm = fmaxf(fmaxf(gm, fmaxf(fabsf(g.x), fabsf(g.y))), fmaxf(fabsf(g.z), fabsf(g.w)));
        }
      }
    }
  }
  sa[rl][c4l] = a; sb[rl][c4l] = b;
  if (gmaxp) { gm = wave_max(gm); if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = gm; }
  __syncthreads();
  if (gmaxp && threadIdx.x == 0) gmaxp[(long long)blockIdx.y * gridDim.x + blockIdx.x] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
  if (rl == 0 && c < C) {
#pragma unroll
    for (int i = 1; i < 16; ++i) {
      a.x += sa[i][c4l].x; a.y += sa[i][c4l].y; a.z += sa[i][c4l].z; a.w += sa[i][c4l].w;
      b.x += sb[i][c4l].x; b.y += sb[i][c4l].y; b.z += sb[i][c4l].z; b.w += sb[i][c4l].w;
    }
    *reinterpret_cast<float4*>(part + ((long long)blockIdx.y * 2 + 0) * C + c) = a;
    *reinterpret_cast<float4*>(part + ((long long)blockIdx.y * 2 + 1) * C + c) = b;
  }
}

__global__ void __launch_bounds__(256) bn_pool_bwd_apply_kernel(const float* __restrict__ dpool,
                                                                 const unsigned char* __restrict__ idx,
                                                                 const float* __restrict__ x, float* __restrict__ dy,
                                                                 long long n4, int C4, PoolGeom pg,
                                                                 const float* __restrict__ gamma, BnBuf bn,
                                                                 const float* __restrict__ k2,
                                                                 const float* __restrict__ k3,
                                                                 unsigned short* __restrict__ hi,
                                                                 unsigned short* __restrict__ mid,
                                                                 unsigned short* __restrict__ lo, const float* __restrict__ f16_slot,
                                                                 FastDiv dC4) {
  const long long stride = (long long)gridDim.x * 256;
  const float fs = f16_slot ? f16_slot[0] : 1.f;      // f16x2 planes (lo == NULL): the device-resident scale of this gradient
  // C4 divides 256 (checked by the host) and the grid stride is a multiple of 256: a thread stays on one channel quad, whose nine
  // per-channel constants are read once; the element's row is a 32-bit quotient
  const int c4 = (int)(threadIdx.x % (unsigned)C4), c = c4 * 4;
  const float4 sc = *reinterpret_cast<const float4*>(bn.scale + c);
  const float4 sh = *reinterpret_cast<const float4*>(bn.shift + c);
  const float4 mu = *reinterpret_cast<const float4*>(bn.mean + c);
  const float4 is = *reinterpret_cast<const float4*>(bn.invstd + c);
  const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
  const float4 a2 = *reinterpret_cast<const float4*>(k2 + c);
  const float4 a3 = *reinterpret_cast<const float4*>(k3 + c);
  auto finish = [&](long long i, const float4 v) {
    const unsigned row = fast_div((unsigned)i, dC4);
    const float4 g = pool_relu_grad4(dpool, idx, v, sc, sh, row, c4, C4, pg);
    float4 o;
    o.x = ga.x * is.x * (g.x - a2.x - (v.x - mu.x) * is.x * a3.x);
    o.y = ga.y * is.y * (g.y - a2.y - (v.y - mu.y) * is.y * a3.y);
    o.z = ga.z * is.z * (g.z - a2.z - (v.z - mu.z) * is.z * a3.z);
    o.w = ga.w * is.w * (g.w - a2.w - (v.w - mu.w) * is.w * a3.w);
    reinterpret_cast<float4*>(dy)[i] = o;
    if (hi) {     // the same gradient as paired bf16x3 planes (operand of the data-gradient convolution)
      const long long off = plane_offset((long long)row, c, C4 / 8, 1);
      unsigned short h[4], m[4], l[4];
      if (!lo) {      // f16x2 format: two fp16 planes of fs * o
        split2_f16(o.x, fs, h[0], m[0]); split2_f16(o.y, fs, h[1], m[1]); split2_f16(o.z, fs, h[2], m[2]); split2_f16(o.w, fs, h[3], m[3]);
      } else {
        split3_bf16(o.x, h[0], m[0], l[0]); split3_bf16(o.y, h[1], m[1], l[1]);
        split3_bf16(o.z, h[2], m[2], l[2]); split3_bf16(o.w, h[3], m[3], l[3]);
      }
      *reinterpret_cast<uint2*>(hi + off) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
      *reinterpret_cast<uint2*>(mid + off) = make_uint2((unsigned)m[0] | ((unsigned)m[1] << 16), (unsigned)m[2] | ((unsigned)m[3] << 16));
      if (lo) *reinterpret_cast<uint2*>(lo + off) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
    }
  };
  // two elements per trip: both loads of the full-size map are in flight before the first store (175 MB in, 175 MB out at layer 1 of
  // the depth encoder: 3.5 TB/s with one)
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; i + stride < n4; i += 2 * stride) {
    const float4 v0 = reinterpret_cast<const float4*>(x)[i], v1 = reinterpret_cast<const float4*>(x)[i + stride];
    finish(i, v0);
    finish(i + stride, v1);
  }
  if (i < n4) finish(i, reinterpret_cast<const float4*>(x)[i]);
}

int bn_pool_backward(const float* dpool, const unsigned char* idx, const float* x, int B, int H, int W, int C, int k,
                     const float* gamma, BnBuf bn, float* dgamma, float* dbeta, float* ws, float* dy, hipStream_t st,
                     unsigned short* const dy_planes[3], F16Scale* f16) {
  DIC_REQUIRE(C % 64 == 0, "bn_pool_backward: C %% 64");
  const bool f16x2 = dy_planes && !dy_planes[2];
  DIC_REQUIRE(!f16x2 || (f16 && f16->bound && f16->slot), "bn_pool_backward: f16x2 planes need a scale slot");
  const long long rows = (long long)B * H * W;
  DIC_REQUIRE((long long)B * H * W * C / 4 < (1ll << 31) && 256 % (C / 4) == 0, "bn_pool_backward: index range / C <= 1024 with C / 4 dividing 256");
  const PoolGeom pg{H, W, k, H / k, W / k, make_fastdiv((unsigned)H), make_fastdiv((unsigned)W), make_fastdiv((unsigned)k)};
  float* part = ws;
  float* k2 = ws + (size_t)kBnChunksMax * 2 * C;
  float* k3 = k2 + C;
  float* gmaxp = k3 + C;
  const int chunks = reduce_chunks(C);
  hipLaunchKernelGGL(bn_pool_bwd_reduce_kernel, dim3(C / 64, chunks), dim3(256), 0, st, dpool, idx, x, rows, C, pg, bn,
                     part, f16x2 ? gmaxp : nullptr);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, 32)), dim3(256), 0, st, part, chunks, C, (double)rows,
                     dgamma, dbeta, k2, k3, (const float*)gmaxp, gamma, (const float*)bn.invstd, f16x2 ? f16->bound : nullptr);
  if (f16x2) DIC_TRY(f16_scale_finish(*f16, st));
  const long long n4 = rows * C / 4;
  hipLaunchKernelGGL(bn_pool_bwd_apply_kernel, dim3(ew_blocks(n4)), dim3(256), 0, st, dpool, idx, x, dy, n4, C / 4, pg,
                     gamma, bn, k2, k3, dy_planes ? dy_planes[0] : nullptr, dy_planes ? dy_planes[1] : nullptr,
                     dy_planes ? dy_planes[2] : nullptr, f16x2 ? (const float*)f16->slot : nullptr, make_fastdiv((unsigned)(C / 4)));
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

// column sums with a fixed two-stage tree (deterministic).  Stage 1: block = 64 channels (16 float4 lanes) x 16 row
// lanes over one of 64 row chunks -> ws[chunk][C]; stage 2 sums the 64 chunk rows.  (C % 4 == 0; else scalar path)
__global__ void __launch_bounds__(256) colsum_rows_v4_kernel(const float* __restrict__ X, long long ld, long long rows,
                                                              int C, float* __restrict__ out) {
  __shared__ float4 sa[16][16];
  const int c4l = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + c4l * 4;
  const long long per = (rows + gridDim.y - 1) / gridDim.y;
  const long long r0 = (long long)blockIdx.y * per, r1 = min(rows, r0 + per);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < C)
    for (long long r = r0 + rl; r < r1; r += 128) {       // 8 rows in flight per thread
      float4 g[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long long rr = r + 16 * u;
        g[u] = *reinterpret_cast<const float4*>(X + min(rr, r1 - 1) * ld + c);     // branch-free guard
        if (rr >= r1) g[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { a.x += g[u].x; a.y += g[u].y; a.z += g[u].z; a.w += g[u].w; }
    }
  sa[rl][c4l] = a;
  __syncthreads();
  if (rl == 0 && c < C) {
#pragma unroll
    for (int i = 1; i < 16; ++i) { a.x += sa[i][c4l].x; a.y += sa[i][c4l].y; a.z += sa[i][c4l].z; a.w += sa[i][c4l].w; }
    *reinterpret_cast<float4*>(out + (long long)blockIdx.y * C + c) = a;
  }
}

__global__ void __launch_bounds__(256) colsum_rows_kernel(const float* __restrict__ X, long long ld, long long rows,
                                                           int C, float* __restrict__ out, int chunks) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const long long per = (rows + chunks - 1) / chunks;
  const long long r0 = (long long)blockIdx.y * per, r1 = min(rows, r0 + per);
  float s = 0.f;
#pragma unroll 8
  for (long long r = r0; r < r1; ++r) s += X[r * ld + c];
  out[(long long)blockIdx.y * C + c] = s;
}

int colsum_rows(const float* X, long long ld, long long rows, int C, float* out, float* ws, hipStream_t st) {
  const bool v4 = (C % 4 == 0) && (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0) && rows >= 256;
  if (v4) {
    const int chunks = std::min(256, reduce_chunks(C));   // ws holds up to 256 x C partial rows
    hipLaunchKernelGGL(colsum_rows_v4_kernel, dim3(ceil_div(C, 64), chunks), dim3(256), 0, st, X, ld, rows, C, ws);
    hipLaunchKernelGGL(colsum_rows_kernel, dim3(ceil_div(C, 256), 1), dim3(256), 0, st, ws, (long long)C,
                       (long long)chunks, C, out, 1);
    DIC_LAUNCH_CHECK();
    return DIC_OK;
  }
  const int chunks = (int)std::min<long long>(256, std::max<long long>(1, rows / 16));
  if (chunks > 1) {
    hipLaunchKernelGGL(colsum_rows_kernel, dim3(ceil_div(C, 256), chunks), dim3(256), 0, st, X, ld, rows, C, ws, chunks);
    hipLaunchKernelGGL(colsum_rows_kernel, dim3(ceil_div(C, 256), 1), dim3(256), 0, st, ws, (long long)C,
                       (long long)chunks, C, out, 1);
  } else {
    hipLaunchKernelGGL(colsum_rows_kernel, dim3(ceil_div(C, 256), 1), dim3(256), 0, st, X, ld, rows, C, out, 1);
  }
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

// ------------------------------------------------------------------------------------------
// weight layout transforms
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) oihw_ohwi_kernel(const float* __restrict__ src, float* __restrict__ dst, int O,
                                                         int I, int KH, int KW, int to_ohwi) {
  const long long total = (long long)O * I * KH * KW;
  const long long stride = (long long)gridDim.x * 256;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
    // e enumerates the OHWI order
    const int i = (int)(e % I);
    long long r = e / I;
    const int kw = (int)(r % KW); r /= KW;
    const int kh = (int)(r % KH);
    const int o = (int)(r / KH);
    const long long oihw = (((long long)o * I + i) * KH + kh) * KW + kw;
    if (to_ohwi) dst[e] = src[oihw];
    else dst[oihw] = src[e];
  }
}

int oihw_to_ohwi(const float* src, float* dst, int O, int I, int KH, int KW, hipStream_t st) {
  const long long total = (long long)O * I * KH * KW;
  hipLaunchKernelGGL(oihw_ohwi_kernel, dim3(ew_blocks(total)), dim3(256), 0, st, src, dst, O, I, KH, KW, 1);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

int ohwi_to_oihw(const float* src, float* dst, int O, int I, int KH, int KW, hipStream_t st) {
  const long long total = (long long)O * I * KH * KW;
  hipLaunchKernelGGL(oihw_ohwi_kernel, dim3(ew_blocks(total)), dim3(256), 0, st, src, dst, O, I, KH, KW, 0);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

}  // namespace dic
