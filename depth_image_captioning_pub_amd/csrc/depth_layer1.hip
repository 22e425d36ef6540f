// Backward of the depth encoder's FIRST layer (conv 1 -> 128, k7 s3, + BatchNorm + ReLU + max-pool 3; depth_models.py:19-20,36-39)
// without ever forming the full-size gradient.  Test infrastructure aside, nothing here touches oracle/.
//
// Until round 4 this layer's backward was three passes over its 175-MB pre-pool map x1 = conv1(depth) [B,73,73,128] plus a 175-MB
// gradient written and read back: bn_pool_bwd_reduce (reads x1), bn_pool_bwd_apply (reads x1, writes dy1), conv1_depth_wgrad (reads
// dy1) - 0.24 ms of main-stream kernels and 0.7 GB of HBM streams per step, which is what the step pays for.
// None of it is needed, because this layer is special twice over: its input has ONE channel and is detached (no data gradient), and
// the gradient g that reaches x1 through pool + ReLU is non-zero at one position per pooling window and channel only.  With
//     dy1 = gamma * invstd * (g - k2 - xhat * k3),   k2 = mean(g),  k3 = mean(g * xhat),  xhat = (x1 - mean) * invstd
// the weight gradient dW[c][tap] = sum_pos dy1(pos,c) * patch(pos,tap) splits into
//     gamma_c invstd_c [ A[c][tap]  -  k2_c S[tap]  -  k3_c invstd_c ( (W G)[c][tap] + (b_c - mean_c) S[tap] ) ]
//   A[c][tap] = sum over pooling cells of g_cell(c) * patch(selected position of (cell,c), tap)        (sparse: 36 864 cells)
//   S[tap]    = sum_pos patch(pos,tap),   G[t][tap] = sum_pos patch(pos,t) patch(pos,tap)               (49 + 49x49, channel-free)
// since x1(pos,c) = sum_t W[c][t] patch(pos,t) + b_c.  k2, k3 (and dgamma, dbeta) need g and xhat at the selected positions only.
// Kernels: l1_sparse_reduce (g per cell, partial sums for bn_bwd_finalize), l1_patch_gram (S, G on the exact-fp32 matrix cores: fp32
// inside a workgroup's ~1300 positions, fp64 across workgroups), l1_sparse_wgrad (A), l1_combine (fp64).  The conv bias gradient is exactly zero in front of a
// train-mode BatchNorm (quirk Q10) and is written as such (the dense route produced rounding noise there).
// Accuracy: the dense route summed 341 056 fp32 terms per weight; here the long sums are short or fp64 - the result is closer to the
// fp64 oracle than before (tests/test_encoders_gpu.py, decision-replay tests).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.h"
#include "conv.h"
#include "nn_kernels.h"

namespace dic {

namespace {
constexpr int kC = 128, kKS = 7, kKK = 49;
constexpr int kGN = 64;                 // the Gram matrix is computed as 64 x 64: taps 0..48, "tap" 49 = the constant 1 (its row / column = S), rest zero
constexpr int kReduceChunks = 512, kGramBlocks = 256, kWgradBlocks = 768;
constexpr int kCellsPerThread = 9;      // l1_sparse_reduce: cells a thread gathers in one batch (512 chunks x 8 lanes x 9 = 36 864 cells at the bench shape)

struct L1Geom { int B, H, W, H1, W1, PH, PW; };      // input map, conv output, pooled map (k = s = 3)

// ---- g per (cell, channel) and the partial sums of BatchNorm backward: part[chunk][0][c] = sum g, [1][c] = sum g * xhat.
// xsel = conv1's raw output at each window's argmax, left by the forward's max-pool (gathering it from the 175-MB map here would
// touch every 128-B line of it: the first version of this kernel moved 170 MB and took 43 us).
__global__ void __launch_bounds__(256) l1_sparse_reduce_kernel(const float* __restrict__ dpool, const float* __restrict__ xsel, int ncells,
                                                               BnBuf bn, float* __restrict__ part, float* __restrict__ gsel, int per) {
  __shared__ float4 sa[8][32], sb[8][32];
  const int c4 = threadIdx.x & 31, cl = threadIdx.x >> 5, c = c4 * 4;
  const int r0 = blockIdx.x * per, r1 = min(ncells, r0 + per);
  const float4 mu = *reinterpret_cast<const float4*>(bn.mean + c), is = *reinterpret_cast<const float4*>(bn.invstd + c);
  const float4 sc = *reinterpret_cast<const float4*>(bn.scale + c), sh = *reinterpret_cast<const float4*>(bn.shift + c);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  for (int cell0 = r0 + cl; cell0 < r1; cell0 += 8 * kCellsPerThread) {
    float4 d[kCellsPerThread], xv[kCellsPerThread];
#pragma unroll
    for (int u = 0; u < kCellsPerThread; ++u) {
      const int cell = min(cell0 + 8 * u, r1 - 1);                       // (branch-free guard: re-read the last cell, masked below)
      d[u] = reinterpret_cast<const float4*>(dpool)[(long long)cell * 32 + c4];
      xv[u] = reinterpret_cast<const float4*>(xsel)[(long long)cell * 32 + c4];
    }
#pragma unroll
    for (int u = 0; u < kCellsPerThread; ++u) {
      const int cell = cell0 + 8 * u;
      if (cell < r1) {      // the gradient reaches the window's argmax where the BatchNorm output is positive (pool_relu_grad4, nn_kernels.hip)
        float4 gg;
        gg.x = (xv[u].x * sc.x + sh.x > 0.f) ? d[u].x : 0.f;
        gg.y = (xv[u].y * sc.y + sh.y > 0.f) ? d[u].y : 0.f;
        gg.z = (xv[u].z * sc.z + sh.z > 0.f) ? d[u].z : 0.f;
        gg.w = (xv[u].w * sc.w + sh.w > 0.f) ? d[u].w : 0.f;
        reinterpret_cast<float4*>(gsel)[(long long)cell * 32 + c4] = gg;
        a.x += gg.x; a.y += gg.y; a.z += gg.z; a.w += gg.w;
        b.x += gg.x * (xv[u].x - mu.x) * is.x; b.y += gg.y * (xv[u].y - mu.y) * is.y;
        b.z += gg.z * (xv[u].z - mu.z) * is.z; b.w += gg.w * (xv[u].w - mu.w) * is.w;
      }
    }
  }
  sa[cl][c4] = a; sb[cl][c4] = b;
  __syncthreads();
  if (cl == 0) {
#pragma unroll
    for (int i = 1; i < 8; ++i) {
      a.x += sa[i][c4].x; a.y += sa[i][c4].y; a.z += sa[i][c4].z; a.w += sa[i][c4].w;
      b.x += sb[i][c4].x; b.y += sb[i][c4].y; b.z += sb[i][c4].z; b.w += sb[i][c4].w;
    }
    *reinterpret_cast<float4*>(part + ((long long)blockIdx.x * 2 + 0) * kC + c) = a;
    *reinterpret_cast<float4*>(part + ((long long)blockIdx.x * 2 + 1) * kC + c) = b;
  }
}

// ---- the 64 x 64 Gram matrix of the patches ("tap" 49 = 1, so row 49 holds S) on the exact-fp32 matrix cores: per pair of output
// positions one v_mfma_f32_32x32x2_f32 per 32 x 32 block, each lane fetching its own operand element from the seven input rows staged
// in LDS (row stride W + 1: the rows of a column fall into different banks).  Wave w of the four owns block (w >> 1, w & 1) with two
// accumulators (even / odd position pairs: two independent chains).  fp32 sums over a workgroup's ~1300 positions, written as floats.
__global__ void __launch_bounds__(256) l1_patch_gram_kernel(const float* __restrict__ depth, L1Geom g, float* __restrict__ out, int seg_rows,
                                                            int Wp) {
  // LDS: [3 * seg_rows + 4][Wp] input rows of a run of seg_rows output rows of one image (Wp >= 3 * (W1 + 3) + 7, odd: columns past W are
  // zero, so positions past the row end multiply zeros), then one row of ones at the columns 3 p, p < W1 ("tap" 49) and one row of zeros
  // (taps 50..63): every operand is ONE LDS read at (lane base + 12 floats per step), no selects in the loop.
  extern __shared__ float rows[];
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i31 = lane & 31, kk = lane >> 5;
  const int nin_max = 3 * seg_rows + 4;
  float* const ones = rows + nin_max * Wp;
  float* const zeros = ones + Wp;
  for (int e = tid; e < Wp; e += 256) { ones[e] = (e % 3 == 0 && e < 3 * g.W1) ? 1.f : 0.f; zeros[e] = 0.f; }
  const int tapA = (w >> 1) * 32 + i31, tapB = (w & 1) * 32 + i31;
  // per-lane operand base (floats from `rows`) for output row j = 0 and position kk; row-independent for the ones / zeros taps
  const int baseA = tapA < kKK ? (tapA / kKS) * Wp + (tapA % kKS) + 3 * kk : (tapA == kKK ? nin_max * Wp : (nin_max + 1) * Wp) + 3 * kk;
  const int baseB = tapB < kKK ? (tapB / kKS) * Wp + (tapB % kKS) + 3 * kk : (tapB == kKK ? nin_max * Wp : (nin_max + 1) * Wp) + 3 * kk;
  const int stepA = tapA < kKK ? 3 * Wp : 0, stepB = tapB < kKK ? 3 * Wp : 0;      // per output row
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  const int rows_total = g.B * g.H1;
  const int per = (rows_total + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * per, r1 = min(rows_total, r0 + per);
  const int steps_per_row = (g.W1 + 3) / 4;
  for (int rs = r0; rs < r1;) {        // segments: runs of <= seg_rows output rows inside one image, staged with one round trip
    const int b = rs / g.H1, oh0 = rs - b * g.H1;
    const int n = min(min(seg_rows, r1 - rs), g.H1 - oh0);
    const int nin = 3 * n + 4;                           // input rows 3 * oh0 .. 3 * (oh0 + n - 1) + 6
    const float* src = depth + ((long long)b * g.H + (long long)oh0 * 3) * g.W;
    __syncthreads();
    for (int e0 = tid; e0 < nin * Wp; e0 += 256 * 8) {      // eight loads in flight per thread, then eight LDS stores
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + 256 * u, rr = e / Wp, cc = e - rr * Wp;
        v[u] = (e < nin * Wp && cc < g.W) ? src[rr * g.W + cc] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int e = e0 + 256 * u; if (e < nin * Wp) rows[e] = v[u]; }
    }
    __syncthreads();
    for (int j = 0; j < n; ++j) {
      const float* pa = rows + baseA + j * stepA;
      const float* pb = rows + baseB + j * stepB;
      // positions 4 s + kk (acc0) and 4 s + 2 + kk (acc1); the operands of step s + 1 are read before the MFMAs of step s are issued
      float a0 = pa[0], b0 = pb[0], a1 = pa[6], b1 = pb[6];
      for (int sidx = 0; sidx + 1 < steps_per_row; ++sidx) {
        pa += 12; pb += 12;
        const float na0 = pa[0], nb0 = pb[0], na1 = pa[6], nb1 = pb[6];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc1, 0, 0, 0);
        a0 = na0; b0 = nb0; a1 = na1; b1 = nb1;
      }
      {   // last step of the row: positions past W1 - 1 would read real pixels (their windows overlap the last valid one) - masked
        const int p0 = 4 * (steps_per_row - 1) + kk;
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(p0 < g.W1 ? a0 : 0.f, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(p0 + 2 < g.W1 ? a1 : 0.f, b1, acc1, 0, 0, 0);
      }
    }
    rs += n;
  }
  float* o = out + (long long)blockIdx.x * (kGN * kGN);
#pragma unroll
  for (int r = 0; r < 16; ++r) {       // C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int row = (w >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk, col = (w & 1) * 32 + i31;
    o[row * kGN + col] = acc0[r] + acc1[r];
  }
}

// the per-workgroup float partials summed in fp64: 64 entries x 4 partial lanes per workgroup, lanes then entries in fixed order
__global__ void __launch_bounds__(256) l1_gram_reduce_kernel(const float* __restrict__ partial, int nblk, double* __restrict__ gs) {
  __shared__ double sp[4][64];
  const int el = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;
  double s = 0.0;
  for (int w0 = pl; w0 < nblk; w0 += 32) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int w = w0 + 4 * u; v[u] = w < nblk ? partial[(long long)w * (kGN * kGN) + e] : 0.f; }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (double)v[u];
  }
  sp[pl][el] = s;
  __syncthreads();
  if (pl == 0) gs[e] = (sp[0][el] + sp[1][el]) + (sp[2][el] + sp[3][el]);
}

// ---- A[c][tap] = sum over cells of g_cell(c) * patch(selected position, tap).  Workgroup = a run of whole pooled rows; the 13 input
// rows under a pooled row are staged in LDS once, then its cells go two at a time (one per half of the 256 threads, thread =
// channel), the next cell's gradient / index already loaded while this one's 49 products run.
template <int Wp>      // LDS row stride, compile-time (the 49 reads of a cell carry immediate offsets) and = 3 mod 32: the nine window positions a
                       // wave's lanes read from start at 3 ih Wp + 3 iw = 9 ih + 3 iw mod 32 - nine different banks (with stride = 1 mod 32 three of
                       // them collided and the kernel ran at a third of the LDS rate: 48 us)
__global__ void __launch_bounds__(256) l1_sparse_wgrad_kernel(const float* __restrict__ depth, const float* __restrict__ gsel,
                                                              const unsigned char* __restrict__ idx, L1Geom g,
                                                              float* __restrict__ ws) {      // ws[block][128 * 49]
  extern __shared__ float lds[];       // [13][Wp] | red[128 * 49]
  float* const reg = lds;
  float* const red = lds + 13 * Wp;
  const int tid = threadIdx.x, c = tid & 127, half = tid >> 7;
  const int prow_total = g.B * g.PH;
  const int per = (prow_total + gridDim.x - 1) / gridDim.x;
  const int q0 = blockIdx.x * per, q1 = min(prow_total, q0 + per);
  float acc[kKK];
#pragma unroll
  for (int k = 0; k < kKK; ++k) acc[k] = 0.f;
  // the 13 input rows of pooled row q+1 are loaded into registers before the cells of row q run, and stored to LDS after them
  constexpr int NPF = (13 * (Wp - 3) + 255) / 256;      // loads per thread for a 13 x (Wp - 3) region (W <= Wp - 3)
  float pf[NPF];
  auto row_src = [&](int q) { const int bi = q / g.PH, ph = q - bi * g.PH; return depth + ((long long)bi * g.H + 9 * ph) * g.W; };
  auto prefetch_rows = [&](int q) {
    const float* src = row_src(q);
#pragma unroll
    for (int u = 0; u < NPF; ++u) { const int e = tid + 256 * u; pf[u] = e < 13 * g.W ? src[e] : 0.f; }
  };
  auto store_rows = [&]() {
#pragma unroll
    for (int u = 0; u < NPF; ++u) { const int e = tid + 256 * u; if (e < 13 * g.W) { const int rr = e / g.W; reg[rr * Wp + (e - rr * g.W)] = pf[u]; } }
  };
  if (q0 < q1) prefetch_rows(q0);
  for (int q = q0; q < q1; ++q) {
    __syncthreads();                   // everyone is done with the previous row's image
    store_rows();
    const long long cell0 = (long long)q * g.PW;
    float gv = half < g.PW ? gsel[(cell0 + half) * kC + c] : 0.f;      // (issued before the row prefetch: loads retire in order)
    int id = half < g.PW ? idx[(cell0 + half) * kC + c] : 0;
    if (q + 1 < q1) prefetch_rows(q + 1);
    __syncthreads();
    for (int pw = half; pw < g.PW; pw += 2) {
      const float gcur = gv;
      const int icur = id;
      if (pw + 2 < g.PW) { gv = gsel[(cell0 + pw + 2) * kC + c]; id = idx[(cell0 + pw + 2) * kC + c]; }
      const int ih = icur / 3, iw = icur - 3 * ih;
      const float* p = reg + (3 * ih) * Wp + 9 * pw + 3 * iw;
#pragma unroll
      for (int kh = 0; kh < kKS; ++kh)
#pragma unroll
        for (int kw = 0; kw < kKS; ++kw) acc[kh * kKS + kw] = fmaf(gcur, p[kh * Wp + kw], acc[kh * kKS + kw]);
    }
  }
  __syncthreads();
  if (half == 1) {
#pragma unroll
    for (int k = 0; k < kKK; ++k) red[c * kKK + k] = acc[k];
  }
  __syncthreads();
  if (half == 0) {
#pragma unroll
    for (int k = 0; k < kKK; ++k) red[c * kKK + k] += acc[k];
  }
  __syncthreads();
  float4* const o = reinterpret_cast<float4*>(ws + (long long)blockIdx.x * (kC * kKK));      // (coalesced: a thread-per-channel store is 196 B apart per lane)
  for (int e = tid; e < kC * kKK / 4; e += 256) o[e] = reinterpret_cast<const float4*>(red)[e];
}

// ---- dW[c][tap] (OIHW, C_in = 1) in fp64 from A, S, G, the weights and the BatchNorm quantities; db = 0
__global__ void __launch_bounds__(256) l1_combine_kernel(const float* __restrict__ A, const double* __restrict__ gs,
                                                         const float* __restrict__ W, const float* __restrict__ bias,
                                                         const float* __restrict__ gamma, BnBuf bn, const float* __restrict__ k2,
                                                         const float* __restrict__ k3, float* __restrict__ dW,
                                                         float* __restrict__ db) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < kC && db) db[e] = 0.f;
  if (e >= kC * kKK) return;
  const int c = e / kKK, tap = e - c * kKK;
  double wg = 0.0;
  for (int t = 0; t < kKK; ++t) wg += (double)W[c * kKK + t] * gs[t * kGN + tap];
  const double S = gs[kKK * kGN + tap];
  const double is = (double)bn.invstd[c];
  const double v = (double)A[e] - (double)k2[c] * S - (double)k3[c] * is * (wg + ((double)bias[c] - (double)bn.mean[c]) * S);
  dW[e] = (float)((double)gamma[c] * is * v);
}
}  // namespace

bool depth_layer1_sparse_supported(const ConvDesc& d) {
  return d.C == 1 && d.CO == kC && d.KH == kKS && d.KW == kKS && d.stride == 3 && d.pad == 0 && d.OH() >= 3 && d.OW() >= 3 &&
         d.W <= 640;
}
size_t depth_layer1_sparse_ws_floats(const ConvDesc& d) {      // gsel | gram partials | gs (doubles) | A | wgrad partials
  const size_t cells = (size_t)d.B * (d.OH() / 3) * (d.OW() / 3);
  return cells * kC + (size_t)kGramBlocks * kGN * kGN + 2 * (size_t)kGN * kGN + 8 + (size_t)kC * kKK + (size_t)kWgradBlocks * kC * kKK;
}

// dpool / idx: gradient and argmax of the pooled map [B, OH/3, OW/3, 128]; xsel: conv1's raw output at those argmax positions
// (bn_relu_maxpool's optional output); bn: the layer's forward BatchNorm
// (scale, shift, mean, invstd); bn_ws: >= bn_backward_ws_floats(128); ws: >= depth_layer1_sparse_ws_floats(d); cs_ws: colsum scratch.
int depth_layer1_backward_sparse(const float* depth, const ConvDesc& d, const float* dpool, const unsigned char* idx, const float* xsel,
                                 const float* conv_w, const float* conv_b, const float* gamma, BnBuf bn, float* dgamma, float* dbeta,
                                 float* dW, float* db, float* bn_ws, float* ws, float* cs_ws, hipStream_t st) {
  DIC_REQUIRE(depth_layer1_sparse_supported(d), "depth_layer1_backward_sparse: expects the 1 -> 128 channel 7x7 stride-3 convolution");
  const L1Geom g{d.B, d.H, d.W, d.OH(), d.OW(), d.OH() / 3, d.OW() / 3};
  const size_t cells = (size_t)g.B * g.PH * g.PW;
  float* gsel = ws;
  float* gram_part = gsel + cells * kC;
  float* after = gram_part + (size_t)kGramBlocks * kGN * kGN;
  if (reinterpret_cast<uintptr_t>(after) & 7) ++after;      // (8-byte alignment of the fp64 sums; the size leaves room)
  double* gs = reinterpret_cast<double*>(after);
  float* A = reinterpret_cast<float*>(gs + kGN * kGN);
  float* wpart = A + kC * kKK;
  float *part, *k2, *k3;
  bn_backward_ws_layout(bn_ws, kC, &part, &k2, &k3);
  // chunks of whole 8 x kCellsPerThread-cell batches, at most kReduceChunks of them
  const int batch = 8 * kCellsPerThread;
  const int per = (int)(((cells + kReduceChunks - 1) / kReduceChunks + batch - 1) / batch * batch);
  const int chunks = (int)((cells + per - 1) / per);
  hipLaunchKernelGGL(l1_sparse_reduce_kernel, dim3(chunks), dim3(256), 0, st, dpool, xsel, (int)cells, bn, part, gsel, per);
  DIC_TRY(bn_backward_finalize(part, chunks, kC, (double)d.M(), dgamma, dbeta, k2, k3, st));
  // Gram: as many output rows per staging as 60 KB of LDS hold (19 at W = 224: a workgroup's whole run)
  const int gsteps = (g.W1 + 3) / 4;                                 // position steps per output row (4 positions each)
  const int gWp = std::max(g.W, 12 * gsteps + 4) | 1;                // LDS row stride: holds column 3 * (4 * steps - 1) + 6 of the last, padded step; odd
  const int seg_rows = std::max(1, std::min(32, (int)((60 * 1024 / sizeof(float) / gWp - 7) / 3)));
  hipLaunchKernelGGL(l1_patch_gram_kernel, dim3(kGramBlocks), dim3(256), (size_t)(3 * seg_rows + 7) * gWp * sizeof(float), st, depth, g,      // (+1 row: the read-ahead of the last step)
                     gram_part, seg_rows, gWp);
  hipLaunchKernelGGL(l1_gram_reduce_kernel, dim3(kGN * kGN / 64), dim3(256), 0, st, (const float*)gram_part, kGramBlocks, gs);
  if (g.W <= 256)
    hipLaunchKernelGGL(l1_sparse_wgrad_kernel<259>, dim3(kWgradBlocks), dim3(256), (size_t)(13 * 259 + kC * kKK) * sizeof(float), st, depth,
                       (const float*)gsel, idx, g, wpart);
  else
    hipLaunchKernelGGL(l1_sparse_wgrad_kernel<643>, dim3(kWgradBlocks), dim3(256), (size_t)(13 * 643 + kC * kKK) * sizeof(float), st, depth,
                       (const float*)gsel, idx, g, wpart);
  DIC_LAUNCH_CHECK();
  DIC_TRY(colsum_rows(wpart, kC * kKK, kWgradBlocks, kC * kKK, A, cs_ws, st));
  hipLaunchKernelGGL(l1_combine_kernel, dim3((kC * kKK + 255) / 256), dim3(256), 0, st, (const float*)A, (const double*)gs, conv_w, conv_b,
                     gamma, bn, (const float*)k2, (const float*)k3, dW, db);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

}  // namespace dic
