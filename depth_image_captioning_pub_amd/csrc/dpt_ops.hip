// Building blocks of the frozen DPT-Hybrid depth front-end (BASELINE config 5):
//   Captioning_models/Depth_caption_model/DPT_model.py:16-67, modules/midas/dpt_depth.py:26-107,
//   modules/midas/blocks.py:231-341, modules/midas/vit.py:36-155,345-477 and - un-vendored, restated from its published
//   definition - timm 0.4.12's vit_base_resnet50_384 (ResNetV2 stem with StdConv2dSame + GroupNorm, ViT-B/16 blocks).
// Forward only (the reference runs it under @torch.no_grad, frozen).  The contractions (convolutions, linear layers) run
// on the split-bf16 kernels of gemm_bf3.hip through dic_conv2d_bf16x3 / dic_linear_bf16x3 (DptRunner's default arithmetic;
// the 3-channel stem and DptRunner(arith="fp32") use the exact-fp32 MFMA kernels of gemm.hip: dic_conv2d_fwd / dic_gemm_f32);
// this file holds what those do not cover: weight standardisation, asymmetric 'SAME' padding, GroupNorm, LayerNorm,
// multi-head attention (matrix cores, split-bf16; round 3), the bilinear x2 up-sampling of the fusion blocks and small
// element-wise passes.  All activations NHWC / [tokens][channels], fp32.
#include "dic.h"
#include "common.h"
#include "nn_kernels.h"

namespace dic {

static inline int dpt_blocks(long long n, int per = 256) { return (int)std::min<long long>((n + per - 1) / per, 65535 * 16); }

// ---- StdConv2d(Same): w_hat = (w - mean) / (std + eps), statistics per output filter, biased std ----------------
__global__ void __launch_bounds__(256) weight_std_kernel(const float* __restrict__ w, int K, float eps,
                                                          float* __restrict__ out) {
  __shared__ double red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* wo = w + (long long)blockIdx.x * K;
  double s = 0.0, s2 = 0.0;
  for (int k = tid; k < K; k += 256) { const double v = wo[k]; s += v; s2 += v * v; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); s2 += __shfl_xor(s2, o, 64); }
  if (lane == 0) { red[0][wv] = s; red[1][wv] = s2; }
  __syncthreads();
  s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  s2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  const double mean = s / K;
  const double var = fmax(s2 / K - mean * mean, 0.0);
  const double inv = 1.0 / (sqrt(var) + (double)eps);
  for (int k = tid; k < K; k += 256) out[(long long)blockIdx.x * K + k] = (float)(((double)wo[k] - mean) * inv);
}

// ---- constant padding of an NHWC tensor (asymmetric 'SAME' padding of StdConv2dSame / MaxPool2dSame) ------------
__global__ void __launch_bounds__(256) pad_nhwc_kernel(const float* __restrict__ x, int H, int W, int C, int top, int left,
                                                        int OH, int OW, float value, long long total,
                                                        float* __restrict__ out) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % C);
    long long r = i / C;
    const int ow = (int)(r % OW); r /= OW;
    const int oh = (int)(r % OH);
    const long long b = r / OH;
    const int h = oh - top, w = ow - left;
    out[i] = ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) ? x[((b * H + h) * W + w) * C + c] : value;
  }
}

// ---- GroupNorm over NHWC (+ optional residual add and ReLU) ---------------------------------------------------------
// Statistics: grid (splits, B*G) - workgroup (s, bg) accumulates sum and sum of squares (fp64) over its slice of the
// group's pixels (float4 loads of the group's contiguous channels) -> part[bg][s][2]; the apply kernel (same grid) first
// combines the `splits` partials in fixed order, then normalises its slice.  Two passes over the data instead of three,
// and enough workgroups to fill the chip when B*G is small (B = 1: 32 groups).
template <int VEC>
__global__ void __launch_bounds__(256) groupnorm_stats_kernel(const float* __restrict__ x, long long HW, int C, int G,
                                                               double* __restrict__ part) {
  __shared__ double red[2][4];
  const int bg = blockIdx.y, b = bg / G, g = bg % G, cg = C / G, v4 = cg / VEC;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long long per = (HW + gridDim.x - 1) / gridDim.x;
  const long long p0 = (long long)blockIdx.x * per, p1 = min(HW, p0 + per);
  const float* base = x + (long long)b * HW * C + (long long)g * cg;
  double s = 0.0, s2 = 0.0;
  const long long n4 = (p1 - p0) * v4;
  for (long long i = tid; i < n4; i += 256) {
    const long long pix = p0 + i / v4;
    const float* px = base + pix * C + (i % v4) * VEC;
    if constexpr (VEC == 4) {
      const float4 v = *reinterpret_cast<const float4*>(px);
      s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
      s2 += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    } else {
      const double v = px[0];
      s += v; s2 += v * v;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); s2 += __shfl_xor(s2, o, 64); }
  if (lane == 0) { red[0][wv] = s; red[1][wv] = s2; }
  __syncthreads();
  if (tid == 0) {
    double* o = part + ((long long)bg * gridDim.x + blockIdx.x) * 2;
    o[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    o[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

template <int VEC>
__global__ void __launch_bounds__(256) groupnorm_apply_kernel(const float* __restrict__ x, long long HW, int C, int G,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps,
                                                               const float* __restrict__ residual, int relu,
                                                               const double* __restrict__ part, float* __restrict__ out) {
  const int bg = blockIdx.y, b = bg / G, g = bg % G, cg = C / G, v4 = cg / VEC;
  const int tid = threadIdx.x;
  double s = 0.0, s2 = 0.0;
  for (int i = 0; i < (int)gridDim.x; ++i) { s += part[((long long)bg * gridDim.x + i) * 2]; s2 += part[((long long)bg * gridDim.x + i) * 2 + 1]; }
  const double n = (double)HW * cg;
  const double mean_d = s / n;
  const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(fmax(s2 / n - mean_d * mean_d, 0.0) + (double)eps));
  const long long per = (HW + gridDim.x - 1) / gridDim.x;
  const long long p0 = (long long)blockIdx.x * per, p1 = min(HW, p0 + per);
  const long long goff = (long long)b * HW * C + (long long)g * cg;
  const long long n4 = (p1 - p0) * v4;
  for (long long i = tid; i < n4; i += 256) {
    const int c4 = (int)(i % v4) * VEC;
    const long long o = goff + (p0 + i / v4) * C + c4;
    if constexpr (VEC == 1) {
      float y = (x[o] - mean) * rstd * gamma[g * cg + c4] + beta[g * cg + c4];
      if (residual) y += residual[o];
      if (relu) y = fmaxf(y, 0.f);
      out[o] = y;
      continue;
    }
    const float4 v = *reinterpret_cast<const float4*>(x + o);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + g * cg + c4), be = *reinterpret_cast<const float4*>(beta + g * cg + c4);
    float4 y;
    y.x = (v.x - mean) * rstd * ga.x + be.x; y.y = (v.y - mean) * rstd * ga.y + be.y;
    y.z = (v.z - mean) * rstd * ga.z + be.z; y.w = (v.w - mean) * rstd * ga.w + be.w;
    if (residual) { const float4 r = *reinterpret_cast<const float4*>(residual + o); y.x += r.x; y.y += r.y; y.z += r.z; y.w += r.w; }
    if (relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
    *reinterpret_cast<float4*>(out + o) = y;
  }
}

// ---- LayerNorm over the last dimension: one wave per row ---------------------------------------------------------
__global__ void __launch_bounds__(256) layernorm_kernel(const float* __restrict__ x, long long rows, int C,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float eps, float* __restrict__ out) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* xr = x + row * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += xr[c];
  const float mean = wave_sum(s) / (float)C;
  float v = 0.f;
  for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; v += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)C + eps);
  for (int c = lane; c < C; c += 64) out[row * C + c] = (xr[c] - mean) * rstd * gamma[c] + beta[c];
}

// ---- multi-head self-attention of a ViT block (timm Attention.forward): softmax(q k^T / sqrt(hd)) v ---------------
// qkv: [B, N, 3, heads, 64] (the layout nn.Linear(dim, 3*dim) + reshape(B, N, 3, heads, hd) gives), out: [B, N, heads*64].
// Workgroup = 64 queries of one (image, head); keys / values stream through LDS in tiles of 64 with an online softmax;
// thread (ty, tx) owns a 4 x 4 block of the 64 x 64 score tile and of the 64 x 64 output tile (plain fp32 FMAs).
constexpr int kHd = 64, kAt = 64, kAld = 68;        // head dim, tile, LDS row stride (16-B aligned, conflict-light)
__global__ void __launch_bounds__(256) vit_attention_kernel(const float* __restrict__ qkv, int N, int heads, float scale,
                                                             float* __restrict__ out) {
  __shared__ __align__(16) float Qs[kAt][kAld], Ks[kAt][kAld], Vs[kAt][kAld], Ps[kAt][kAld];
  const int qt = blockIdx.x, hh = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const long long row_stride = (long long)3 * heads * kHd;
  const float* base = qkv + (long long)b * N * row_stride + (long long)hh * kHd;
  for (int i = tid; i < kAt * (kHd / 4); i += 256) {       // Q tile, pre-scaled
    const int r = i / (kHd / 4), c4 = i % (kHd / 4);
    const int q = qt * kAt + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < N) v = *reinterpret_cast<const float4*>(base + (long long)q * row_stride + c4 * 4);
    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
    *reinterpret_cast<float4*>(&Qs[r][c4 * 4]) = v;
  }
  float m[4], l[4], o[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { m[i] = -INFINITY; l[i] = 0.f; for (int j = 0; j < 4; ++j) o[i][j] = 0.f; }
  const int ntiles = (N + kAt - 1) / kAt;
  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();                                        // previous tile's Ks / Vs / Ps are no longer read
    for (int i = tid; i < kAt * (kHd / 4); i += 256) {
      const int r = i / (kHd / 4), c4 = i % (kHd / 4);
      const int k = kt * kAt + r;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (k < N) {
        const float* p = base + (long long)k * row_stride + c4 * 4;
        kv = *reinterpret_cast<const float4*>(p + (long long)heads * kHd);
        vv = *reinterpret_cast<const float4*>(p + (long long)2 * heads * kHd);
      }
      *reinterpret_cast<float4*>(&Ks[r][c4 * 4]) = kv;
      *reinterpret_cast<float4*>(&Vs[r][c4 * 4]) = vv;
    }
    __syncthreads();
    float s[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s[i][j] = 0.f;
#pragma unroll 4
    for (int d = 0; d < kHd; d += 4) {
      float4 q4[4], k4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) q4[i] = *reinterpret_cast<const float4*>(&Qs[ty * 4 + i][d]);
#pragma unroll
      for (int j = 0; j < 4; ++j) k4[j] = *reinterpret_cast<const float4*>(&Ks[tx * 4 + j][d]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          s[i][j] += q4[i].x * k4[j].x + q4[i].y * k4[j].y + q4[i].z * k4[j].z + q4[i].w * k4[j].w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {                            // online softmax of row ty*4+i over this key tile
      float tm = -INFINITY;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (kt * kAt + tx * 4 + j >= N) s[i][j] = -INFINITY;
        tm = fmaxf(tm, s[i][j]);
      }
#pragma unroll
      for (int w = 8; w > 0; w >>= 1) tm = fmaxf(tm, __shfl_xor(tm, w, 64));      // the 16 lanes tx = 0..15 of this row group
      const float mn = fmaxf(m[i], tm);
      const float corr = expf(m[i] - mn);                   // (first tile: exp(-inf) = 0)
      float ps = 0.f;
      float4 p4;
      p4.x = expf(s[i][0] - mn); p4.y = expf(s[i][1] - mn); p4.z = expf(s[i][2] - mn); p4.w = expf(s[i][3] - mn);
      ps = (p4.x + p4.y) + (p4.z + p4.w);
#pragma unroll
      for (int w = 8; w > 0; w >>= 1) ps += __shfl_xor(ps, w, 64);
      l[i] = l[i] * corr + ps;
      m[i] = mn;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[i][j] *= corr;
      *reinterpret_cast<float4*>(&Ps[ty * 4 + i][tx * 4]) = p4;
    }
    __syncthreads();
#pragma unroll 4
    for (int c = 0; c < kAt; c += 4) {                       // O += P V
      float4 p4[4], v4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) p4[i] = *reinterpret_cast<const float4*>(&Ps[ty * 4 + i][c]);
#pragma unroll
      for (int u = 0; u < 4; ++u) v4[u] = *reinterpret_cast<const float4*>(&Vs[c + u][tx * 4]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o[i][0] += p4[i].x * v4[0].x + p4[i].y * v4[1].x + p4[i].z * v4[2].x + p4[i].w * v4[3].x;
        o[i][1] += p4[i].x * v4[0].y + p4[i].y * v4[1].y + p4[i].z * v4[2].y + p4[i].w * v4[3].y;
        o[i][2] += p4[i].x * v4[0].z + p4[i].y * v4[1].z + p4[i].z * v4[2].z + p4[i].w * v4[3].z;
        o[i][3] += p4[i].x * v4[0].w + p4[i].y * v4[1].w + p4[i].z * v4[2].w + p4[i].w * v4[3].w;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = qt * kAt + ty * 4 + i;
    if (q < N) {
      const float inv = 1.0f / l[i];
      *reinterpret_cast<float4*>(out + ((long long)b * N + q) * heads * kHd + hh * kHd + tx * 4) =
          make_float4(o[i][0] * inv, o[i][1] * inv, o[i][2] * inv, o[i][3] * inv);
    }
  }
}

// ---- the same attention on the matrix cores (round 3): split-bf16 arithmetic as in gemm_bf3.hip (hi + mid + lo planes, six
// products per fp32 product, fp32 accumulate), v_mfma_f32_16x16x32_bf16.
//   pre-pass (vit_split_kv_kernel, once per layer): K of every (image, head) as planes Kp[3][Npad][64] (rows = keys) and V
//   TRANSPOSED as planes VTp[3][64][Npad] (rows = head-dim index), zero-padded to Npad = 64 * ceil(N / 64).  Inside a 64-key
//   tile the columns of VTp are permuted: column 32s + 8g + 4h + j holds key 32s + 16h + 4g + j.  Reason below.
//   main kernel: a workgroup = 64 queries of one (image, head), wave w = queries 16w .. 16w+15.  Everything is laid out so
//   that a LANE BELONGS TO ONE QUERY (q = lane & 15) in both products, which makes the online softmax lane-local:
//     S^T = K Q^T : A = K block (16 keys x 32 d, from LDS), B = Q (registers, loaded once)  ->  C: lane (q, g) holds keys 4g+j
//     O^T = V^T P^T: A = V^T block (16 d x 32 keys, from LDS), B = P                         ->  C: lane (q, g) holds d 4g+j
//   The B operand of the second product needs, per lane (q, g) and 32-key step s, eight probabilities of query q as k-slots
//   8g .. 8g+7: exactly the lane's own S^T values of key blocks 2s (j = 0..3) and 2s+1 - no shuffle, no LDS round trip -
//   provided the A operand uses the same slot -> key assignment, which is what the column permutation of VTp does.
constexpr int kAtLd = 72;     // LDS row stride in bf16 (144 B: the 16 rows of a fragment read start in 16 different 16-B bank groups)

__global__ void __launch_bounds__(256) vit_split_kv_kernel(const float* __restrict__ qkv, int N, int heads, int Npad,
                                                            unsigned short* __restrict__ Kp, unsigned short* __restrict__ VTp) {
  __shared__ float vt[64][65];
  const int kt = blockIdx.x, hh = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const long long row_stride = (long long)3 * heads * kHd;
  const float* base = qkv + (long long)b * N * row_stride + (long long)hh * kHd;
  const long long bh = (long long)b * heads + hh;
  const size_t plane = (size_t)Npad * kHd;
  unsigned short* kp = Kp + bh * 3 * plane;
  unsigned short* vp = VTp + bh * 3 * plane;
  {   // thread = (key r, 16 head-dim values): K planes row-major; V into LDS for the transpose
    const int r = tid >> 2, c = (tid & 3) * 16, key = kt * 64 + r;
    float kv[16], vv[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), d = a;
      if (key < N) {
        const float* src = base + (long long)key * row_stride + c + i * 4;
        a = *reinterpret_cast<const float4*>(src + (long long)heads * kHd);
        d = *reinterpret_cast<const float4*>(src + (long long)2 * heads * kHd);
      }
      kv[i * 4] = a.x; kv[i * 4 + 1] = a.y; kv[i * 4 + 2] = a.z; kv[i * 4 + 3] = a.w;
      vv[i * 4] = d.x; vv[i * 4 + 1] = d.y; vv[i * 4 + 2] = d.z; vv[i * 4 + 3] = d.w;
    }
    unsigned short h[16], m[16], l[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { split3_bf16(kv[i], h[i], m[i], l[i]); vt[r][c + i] = vv[i]; }
    unsigned short* dst = kp + (size_t)key * kHd + c;
    auto pack = [](const unsigned short* x, int o) { return (unsigned)x[o] | ((unsigned)x[o + 1] << 16); };
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      *reinterpret_cast<uint4*>(dst + q * 8) = make_uint4(pack(h, q * 8), pack(h, q * 8 + 2), pack(h, q * 8 + 4), pack(h, q * 8 + 6));
      *reinterpret_cast<uint4*>(dst + plane + q * 8) = make_uint4(pack(m, q * 8), pack(m, q * 8 + 2), pack(m, q * 8 + 4), pack(m, q * 8 + 6));
      *reinterpret_cast<uint4*>(dst + 2 * plane + q * 8) = make_uint4(pack(l, q * 8), pack(l, q * 8 + 2), pack(l, q * 8 + 4), pack(l, q * 8 + 6));
    }
  }
  __syncthreads();
  {   // thread = (head-dim index d, 16 permuted columns): V^T planes
    const int d = tid >> 2, p0 = (tid & 3) * 16;
    unsigned short h[16], m[16], l[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int pp = p0 + i, s2 = pp >> 5, g = (pp >> 3) & 3, hsel = (pp >> 2) & 1, j = pp & 3;
      split3_bf16(vt[32 * s2 + 16 * hsel + 4 * g + j][d], h[i], m[i], l[i]);
    }
    unsigned short* dst = vp + (size_t)d * Npad + kt * 64 + p0;
    auto pack = [](const unsigned short* x, int o) { return (unsigned)x[o] | ((unsigned)x[o + 1] << 16); };
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      *reinterpret_cast<uint4*>(dst + q * 8) = make_uint4(pack(h, q * 8), pack(h, q * 8 + 2), pack(h, q * 8 + 4), pack(h, q * 8 + 6));
      *reinterpret_cast<uint4*>(dst + plane + q * 8) = make_uint4(pack(m, q * 8), pack(m, q * 8 + 2), pack(m, q * 8 + 4), pack(m, q * 8 + 6));
      *reinterpret_cast<uint4*>(dst + 2 * plane + q * 8) = make_uint4(pack(l, q * 8), pack(l, q * 8 + 2), pack(l, q * 8 + 4), pack(l, q * 8 + 6));
    }
  }
}

typedef __bf16 dpt_bf16x8 __attribute__((ext_vector_type(8)));
typedef float dpt_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int dpt_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ dpt_f32x4 dpt_mfma6(const dpt_u32x4 a[3], const dpt_u32x4 b[3], dpt_f32x4 c) {
  // small terms first, as in gemm_bf3.hip: al*bh, ah*bl, am*bm, am*bh, ah*bm, ah*bh   (0 = hi, 1 = mid, 2 = lo)
#define DIC_M(A_, B_) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dpt_bf16x8, a[A_]), __builtin_bit_cast(dpt_bf16x8, b[B_]), c, 0, 0, 0);
  DIC_M(2, 0) DIC_M(0, 2) DIC_M(1, 1) DIC_M(1, 0) DIC_M(0, 1) DIC_M(0, 0)
#undef DIC_M
  return c;
}

__global__ void __launch_bounds__(256, 2) vit_attention_mfma_kernel(const float* __restrict__ qkv, int N, int heads, int Npad,
                                                                     float scale, const unsigned short* __restrict__ Kp,
                                                                     const unsigned short* __restrict__ VTp,
                                                                     float* __restrict__ out) {
  __shared__ __align__(16) unsigned short Ks[3][64][kAtLd], Vs[3][64][kAtLd];      // 27.6 KB each
  const int qt = blockIdx.x, hh = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, ql = lane & 15, g = lane >> 4;
  const long long row_stride = (long long)3 * heads * kHd;
  const long long bh = (long long)b * heads + hh;
  const size_t plane = (size_t)Npad * kHd;
  const unsigned short* kp = Kp + bh * 3 * plane;
  const unsigned short* vp = VTp + bh * 3 * plane;
  const int q = qt * 64 + w * 16 + ql;

  dpt_u32x4 qf[2][3];          // Q fragments (B operand of S^T = K Q^T): k-slots 8g .. 8g+7 of d-step s, pre-scaled (exact: 1/8)
  {
    const float* src = qkv + ((long long)b * N + min(q, N - 1)) * row_stride + (long long)hh * kHd;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const float4 x0 = *reinterpret_cast<const float4*>(src + 32 * s2 + 8 * g), x1 = *reinterpret_cast<const float4*>(src + 32 * s2 + 8 * g + 4);
      const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
      unsigned short h[8], m[8], l[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) split3_bf16(q < N ? v[i] * scale : 0.f, h[i], m[i], l[i]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        qf[s2][0][i] = (unsigned)h[2 * i] | ((unsigned)h[2 * i + 1] << 16);
        qf[s2][1][i] = (unsigned)m[2 * i] | ((unsigned)m[2 * i + 1] << 16);
        qf[s2][2][i] = (unsigned)l[2 * i] | ((unsigned)l[2 * i + 1] << 16);
      }
    }
  }
  float mrow = -INFINITY, lrow = 0.f;
  dpt_f32x4 ot[4];              // O^T: d = 16 db + 4 g + j of query q
#pragma unroll
  for (int i = 0; i < 4; ++i) ot[i] = dpt_f32x4{0.f, 0.f, 0.f, 0.f};

  const int ntiles = Npad / 64;
  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();                                        // the previous tile's fragments have been read
    {   // K and V^T tiles, three planes each: 64 rows x 128 B; thread = (row, 32-B quarter), six planes
      const int r = tid >> 2, c = (tid & 3) * 16;
      uint4 x[3][2], y[3][2];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        const unsigned short* ks = kp + pl * plane + (size_t)(kt * 64 + r) * kHd + c;
        const unsigned short* vs = vp + pl * plane + (size_t)r * Npad + kt * 64 + c;
        x[pl][0] = *reinterpret_cast<const uint4*>(ks); x[pl][1] = *reinterpret_cast<const uint4*>(ks + 8);
        y[pl][0] = *reinterpret_cast<const uint4*>(vs); y[pl][1] = *reinterpret_cast<const uint4*>(vs + 8);
      }
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        *reinterpret_cast<uint4*>(&Ks[pl][r][c]) = x[pl][0]; *reinterpret_cast<uint4*>(&Ks[pl][r][c + 8]) = x[pl][1];
        *reinterpret_cast<uint4*>(&Vs[pl][r][c]) = y[pl][0]; *reinterpret_cast<uint4*>(&Vs[pl][r][c + 8]) = y[pl][1];
      }
    }
    __syncthreads();
    // ---- S^T block kb: keys 16 kb + (4 g + j) of this tile, query q
    dpt_f32x4 st[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      st[kb] = dpt_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        dpt_u32x4 ka[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) ka[pl] = *reinterpret_cast<const dpt_u32x4*>(&Ks[pl][kb * 16 + ql][32 * s2 + 8 * g]);
        st[kb] = dpt_mfma6(ka, qf[s2], st[kb]);
      }
    }
    // ---- online softmax of query q over this tile's 64 keys: 16 values in this lane, the rest in lanes q + 16 g'
    float tm = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (kt * 64 + kb * 16 + 4 * g + j >= N) st[kb][j] = -INFINITY;
        tm = fmaxf(tm, st[kb][j]);
      }
    tm = fmaxf(tm, __shfl_xor(tm, 16, 64));
    tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
    const float mn = fmaxf(mrow, tm);
    const float corr = expf(mrow - mn);                     // (first tile: exp(-inf) = 0)
    float ps = 0.f;
    dpt_u32x4 pf[2][3];           // P fragments (B operand of O^T = V^T P^T): k-slot 4 h + j of step s = key block 2 s + h, key 4 g + j
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      unsigned short h[8], m[8], l[8];
#pragma unroll
      for (int hs = 0; hs < 2; ++hs)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float pv = expf(st[2 * s2 + hs][j] - mn);
          ps += pv;
          split3_bf16(pv, h[4 * hs + j], m[4 * hs + j], l[4 * hs + j]);
        }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        pf[s2][0][i] = (unsigned)h[2 * i] | ((unsigned)h[2 * i + 1] << 16);
        pf[s2][1][i] = (unsigned)m[2 * i] | ((unsigned)m[2 * i + 1] << 16);
        pf[s2][2][i] = (unsigned)l[2 * i] | ((unsigned)l[2 * i + 1] << 16);
      }
    }
    ps += __shfl_xor(ps, 16, 64);
    ps += __shfl_xor(ps, 32, 64);
    lrow = lrow * corr + ps;
    mrow = mn;
    // ---- O^T += V^T P^T
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      ot[db][0] *= corr; ot[db][1] *= corr; ot[db][2] *= corr; ot[db][3] *= corr;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        dpt_u32x4 va[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) va[pl] = *reinterpret_cast<const dpt_u32x4*>(&Vs[pl][db * 16 + ql][32 * s2 + 8 * g]);
        ot[db] = dpt_mfma6(va, pf[s2], ot[db]);
      }
    }
  }
  if (q < N) {
    const float inv = 1.0f / lrow;
    float* dst = out + ((long long)b * N + q) * heads * kHd + hh * kHd + 4 * g;
#pragma unroll
    for (int db = 0; db < 4; ++db)
      *reinterpret_cast<float4*>(dst + 16 * db) = make_float4(ot[db][0] * inv, ot[db][1] * inv, ot[db][2] * inv, ot[db][3] * inv);
  }
}

// ---- bilinear x2 up-sampling, align_corners=True, NHWC (FeatureFusionBlock_custom, Interpolate) -------------------
__global__ void __launch_bounds__(256) upsample2x_nhwc_kernel(const float* __restrict__ x, int H, int W, int C4,
                                                               long long total, float* __restrict__ out) {
  const int OH = 2 * H, OW = 2 * W;
  const float rh = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f, rw = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % C4);
    long long r = i / C4;
    const int ow = (int)(r % OW); r /= OW;
    const int oh = (int)(r % OH);
    const long long b = r / OH;
    const float fh = rh * oh, fw = rw * ow;
    const int h0 = (int)fh, w0 = (int)fw;
    const int h1 = min(h0 + 1, H - 1), w1 = min(w0 + 1, W - 1);
    const float ah = fh - h0, aw = fw - w0;
    const float4* p = reinterpret_cast<const float4*>(x) + b * H * W * C4 + c;
    const float4 v00 = p[((long long)h0 * W + w0) * C4], v01 = p[((long long)h0 * W + w1) * C4];
    const float4 v10 = p[((long long)h1 * W + w0) * C4], v11 = p[((long long)h1 * W + w1) * C4];
    float ah1 = 1.f - ah, aw1 = 1.f - aw, ah0 = ah, aw0 = aw;
    // (scalars on purpose: as a float2 {ah, aw} the four products become v_pk_mul_f32 with op_sel:[0,1], the operand-select form
    // that gives wrong results next to other kernels' waves - see build.py)
    asm volatile("" : "+v"(ah1), "+v"(aw1), "+v"(ah0), "+v"(aw0));
    const float w00 = ah1 * aw1, w01 = ah1 * aw0, w10 = ah0 * aw1, w11 = ah0 * aw0;
    float4 y;
    y.x = w00 * v00.x + w01 * v01.x + w10 * v10.x + w11 * v11.x;
    y.y = w00 * v00.y + w01 * v01.y + w10 * v10.y + w11 * v11.y;
    y.z = w00 * v00.z + w01 * v01.z + w10 * v10.z + w11 * v11.z;
    y.w = w00 * v00.w + w01 * v01.w + w10 * v10.w + w11 * v11.w;
    reinterpret_cast<float4*>(out)[i] = y;
  }
}

// ---- element-wise passes: out = act(a + b[i % period]);  act 0 none, 1 ReLU, 3 GELU (exact, erf) ------------------
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
__global__ void __launch_bounds__(256) add_act_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       long long n, long long period, int act, float* __restrict__ out) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float v = a[i];
    if (b) v += b[i % period];
    if (act == 1) v = fmaxf(v, 0.f);
    else if (act == 3) v = gelu_erf(v);
    out[i] = v;
  }
}

// ---- last layer of the depth head: 1x1 convolution to ONE channel + ReLU (dpt_depth.py:96-98) ----------------------
__global__ void __launch_bounds__(256) pointwise_dot_kernel(const float* __restrict__ x, long long rows, int C,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             int relu, float* __restrict__ out) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < rows; r += stride) {
    const float4* p = reinterpret_cast<const float4*>(x + r * C);
    float s = bias ? bias[0] : 0.f;
    for (int c = 0; c < C / 4; ++c) {
      const float4 v = p[c];
      const float4 u = *reinterpret_cast<const float4*>(w + c * 4);
      s += v.x * u.x + v.y * u.y + v.z * u.z + v.w * u.w;
    }
    out[r] = relu ? fmaxf(s, 0.f) : s;
  }
}

}  // namespace dic

using namespace dic;

extern "C" {

int dic_weight_standardize(const float* w, int O, int K, float eps, float* out, void* stream) {
  DIC_REQUIRE(w && out && O > 0 && K > 0, "weight_standardize: bad arguments");
  hipLaunchKernelGGL(weight_std_kernel, dim3(O), dim3(256), 0, (hipStream_t)stream, w, K, eps, out);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

int dic_pad_nhwc(const float* x, int B, int H, int W, int C, int top, int left, int bottom, int right, float value,
                 float* out, void* stream) {
  DIC_REQUIRE(x && out && B > 0 && H > 0 && W > 0 && C > 0 && top >= 0 && left >= 0 && bottom >= 0 && right >= 0,
              "pad_nhwc: bad arguments");
  const int OH = H + top + bottom, OW = W + left + right;
  const long long total = (long long)B * OH * OW * C;
  hipLaunchKernelGGL(pad_nhwc_kernel, dim3(dpt_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, H, W, C, top, left, OH, OW,
                     value, total, out);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

int dic_maxpool_nhwc(const float* x, int B, int H, int W, int C, int k, int stride, float* out, void* stream) {
  DIC_REQUIRE(x && out && B > 0 && C % 4 == 0 && k >= 1 && stride >= 1 && H >= k && W >= k, "maxpool_nhwc: bad arguments");
  return bn_relu_maxpool(x, B, H, W, C, nullptr, 0, k, stride, 0, out, nullptr, (hipStream_t)stream, nullptr);
}

size_t dic_groupnorm_workspace_bytes(int B, int groups) { return (size_t)B * groups * 64 * 2 * sizeof(double); }

int dic_groupnorm_nhwc(const float* x, int B, long long HW, int C, int groups, const float* gamma, const float* beta,
                       float eps, const float* residual, int relu, float* out, void* workspace, void* stream) {
  DIC_REQUIRE(x && out && gamma && beta && workspace && B > 0 && HW > 0 && groups > 0 && C % groups == 0,
              "groupnorm_nhwc: bad arguments");
  // enough pixel slices for >= ~1024 workgroups, at most 64, at least 256 pixels per slice
  int splits = (int)std::min<long long>(64, std::max<long long>(1, std::min<long long>(1024 / std::max(1, B * groups), HW / 256)));
  if ((C / groups) % 4 == 0) {
    hipLaunchKernelGGL(groupnorm_stats_kernel<4>, dim3(splits, B * groups), dim3(256), 0, (hipStream_t)stream, x, HW, C, groups,
                       (double*)workspace);
    hipLaunchKernelGGL(groupnorm_apply_kernel<4>, dim3(splits, B * groups), dim3(256), 0, (hipStream_t)stream, x, HW, C, groups,
                       gamma, beta, eps, residual, relu, (const double*)workspace, out);
  } else {       // (the 64-channel stem: 2 channels per group)
    hipLaunchKernelGGL(groupnorm_stats_kernel<1>, dim3(splits, B * groups), dim3(256), 0, (hipStream_t)stream, x, HW, C, groups,
                       (double*)workspace);
    hipLaunchKernelGGL(groupnorm_apply_kernel<1>, dim3(splits, B * groups), dim3(256), 0, (hipStream_t)stream, x, HW, C, groups,
                       gamma, beta, eps, residual, relu, (const double*)workspace, out);
  }
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

int dic_layernorm(const float* x, long long rows, int C, const float* gamma, const float* beta, float eps, float* out,
                  void* stream) {
  DIC_REQUIRE(x && out && gamma && beta && rows > 0 && C > 0, "layernorm: bad arguments");
  hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, rows, C, gamma,
                     beta, eps, out);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

size_t dic_vit_attention_workspace_bytes(int B, int N, int heads) {
  const size_t npad = (size_t)(N + 63) / 64 * 64;
  return (size_t)B * heads * 2 * 3 * npad * kHd * sizeof(unsigned short);          // K planes + transposed V planes
}

int dic_vit_attention(const float* qkv, int B, int N, int heads, int head_dim, float* out, void* workspace,
                      size_t workspace_bytes, void* stream) {
  DIC_REQUIRE(qkv && out && B > 0 && N > 0 && heads > 0, "vit_attention: bad arguments");
  DIC_REQUIRE(head_dim == kHd, "vit_attention: head dimension must be 64 (ViT-B/16: 768 / 12)");
  const float scale = 1.0f / sqrtf((float)head_dim);
  if (!workspace) {            // no scratch: plain fp32 FMAs on the vector units (round-2 kernel)
    hipLaunchKernelGGL(vit_attention_kernel, dim3((N + kAt - 1) / kAt, heads, B), dim3(256), 0, (hipStream_t)stream, qkv, N,
                       heads, scale, out);
    DIC_LAUNCH_CHECK();
    return DIC_OK;
  }
  DIC_REQUIRE(workspace_bytes >= dic_vit_attention_workspace_bytes(B, N, heads), "vit_attention: workspace too small");
  const int ntile = (N + 63) / 64, npad = ntile * 64;
  unsigned short* Kp = static_cast<unsigned short*>(workspace);
  unsigned short* VTp = Kp + (size_t)B * heads * 3 * npad * kHd;
  hipLaunchKernelGGL(vit_split_kv_kernel, dim3(ntile, heads, B), dim3(256), 0, (hipStream_t)stream, qkv, N, heads, npad, Kp, VTp);
  hipLaunchKernelGGL(vit_attention_mfma_kernel, dim3(ntile, heads, B), dim3(256), 0, (hipStream_t)stream, qkv, N, heads, npad,
                     scale, (const unsigned short*)Kp, (const unsigned short*)VTp, out);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

int dic_upsample2x_bilinear_nhwc(const float* x, int B, int H, int W, int C, float* out, void* stream) {
  DIC_REQUIRE(x && out && B > 0 && H > 0 && W > 0 && C % 4 == 0, "upsample2x: bad arguments");
  const long long total = (long long)B * 4 * H * W * (C / 4);
  hipLaunchKernelGGL(upsample2x_nhwc_kernel, dim3(dpt_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, H, W, C / 4, total,
                     out);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

int dic_add_act(const float* a, const float* b, long long n, long long period, int act, float* out, void* stream) {
  DIC_REQUIRE(a && out && n > 0 && (b == nullptr || period > 0) && (act == 0 || act == 1 || act == 3), "add_act: bad arguments");
  hipLaunchKernelGGL(add_act_kernel, dim3(dpt_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, n, period, act, out);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

int dic_pointwise_dot(const float* x, long long rows, int C, const float* w, const float* bias, int relu, float* out,
                      void* stream) {
  DIC_REQUIRE(x && w && out && rows > 0 && C % 4 == 0, "pointwise_dot: bad arguments");
  hipLaunchKernelGGL(pointwise_dot_kernel, dim3(dpt_blocks(rows)), dim3(256), 0, (hipStream_t)stream, x, rows, C, w, bias, relu,
                     out);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

}  // extern "C"
