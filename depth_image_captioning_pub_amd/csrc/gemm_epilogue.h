// Epilogue shared by the contraction kernels (gemm.hip, gemm_bf3.hip).
#pragma once
#include "gemm.h"

namespace dic {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// the result scale of a launch: the host factor times the (optional) device-resident ones
__device__ __forceinline__ float ep_alpha(const GemmEpilogue& ep) {
  float a = ep.alpha;
  if (ep.alpha_dev[0]) a *= *ep.alpha_dev[0];
  if (ep.alpha_dev[1]) a *= *ep.alpha_dev[1];
  return a;
}

// epilogue for one output element; returns the stored value (BN statistics use it).  alpha = ep_alpha(ep), taken once by the caller
__device__ __forceinline__ float finalize_store(const GemmEpilogue& ep, int m, int n, float v, float alpha) {
  v *= alpha;
  if (ep.bias) v += ep.bias[n];
  if (ep.act == ACT_RELU) v = fmaxf(v, 0.f);
  else if (ep.act == ACT_SIGMOID) v = sigmoidf_(v);
  else if (ep.act == ACT_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  const long long orow = ep.row_map ? ep.row_map[m] : m;
  float* dst = (ep.C2 && n >= ep.nsplit) ? ep.C2 + orow * ep.ldc2 + (n - ep.nsplit) : ep.C + orow * ep.ldc + n;
  if (ep.accumulate) v += *dst;
  *dst = v;
  return v;
}

// ------------------------------------------------------------------------------------------
// shared epilogue: C/D map of the 32x32 MFMA is col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// `sb` is LDS scratch (>= 2*BN floats) that is free once the main loop has ended on a barrier.
// ------------------------------------------------------------------------------------------
template <int BM, int BN, typename P>
__device__ __forceinline__ void gemm_epilogue(const P& p, f32x16 (&acc)[BM / 64][BN / 64], int tm, int tn,
                                              int z, float* sb) {
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, khalf = lane >> 5;
  const int n0 = tn * BN + wn * WN + (lane & 31);
  const int m0 = tm * BM + wm * WM + 4 * khalf;
  float cs[TN], cs2[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) { cs[j] = 0.f; cs2[j] = 0.f; }
  // Fast path (every convolution tile that lies inside the matrix): plain store (+ per-column bias), straight-line
  // code with one row pointer per register group.  The generic per-element path below is ~1700 instructions per tile
  // and dominated the short-K launches (K = 64: 137 us with it, 40 us without any epilogue).
  const bool fast = p.splitk == 1 && !p.ep.row_map && !p.ep.C2 && p.ep.act == ACT_NONE &&
                    (tm + 1) * BM <= p.M && (tn + 1) * BN <= p.N;
  const float alpha = ep_alpha(p.ep);   // (1 except for the f16x2 operand format: a power of two, exact)
  if (fast) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + j * 32;
      const float b = p.ep.bias ? p.ep.bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float* col = p.ep.C + (long long)(m0 + i * 32) * p.ep.ldc + n;
        float old[16];
        if (p.ep.accumulate) {          // C += result: all 16 old values in flight before the first store
#pragma unroll
          for (int r = 0; r < 16; ++r) old[r] = col[(long long)((r & 3) + 8 * (r >> 2)) * p.ep.ldc];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) old[r] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = fmaf(acc[i][j][r], alpha, b) + old[r];
          col[(long long)((r & 3) + 8 * (r >> 2)) * p.ep.ldc] = v;
          cs[j] += v;
          cs2[j] += v * v;
        }
      }
    }
  } else
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + j * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2);
        if (m < p.M && n < p.N) {
          if (p.splitk > 1) {
            p.ws[((long long)z * p.M + m) * p.N + n] = acc[i][j][r];
          } else {
            const float v = finalize_store(p.ep, m, n, acc[i][j][r], alpha);
            cs[j] += v;
            cs2[j] += v * v;
          }
        }
      }
    }
  if (p.ep.stats && p.splitk == 1) {   // per-(m-tile, column) partial sums for train-mode BatchNorm
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      cs[j] += __shfl_xor(cs[j], 32, 64);
      cs2[j] += __shfl_xor(cs2[j], 32, 64);
    }
    if (wm == 0 && lane < 32) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        sb[wn * WN + j * 32 + lane] = cs[j];
        sb[BN + wn * WN + j * 32 + lane] = cs2[j];
      }
    }
    __syncthreads();
    if (wm == 1 && lane < 32) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + j * 32;
        if (n < p.N) {
          p.ep.stats[((long long)tm * 2 + 0) * p.N + n] = cs[j] + sb[wn * WN + j * 32 + lane];
          p.ep.stats[((long long)tm * 2 + 1) * p.N + n] = cs2[j] + sb[BN + wn * WN + j * 32 + lane];
        }
      }
    }
  }
}

}  // namespace dic
