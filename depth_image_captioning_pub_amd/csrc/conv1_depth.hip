// Depth-encoder layer 1: 7x7 / stride-3 convolution of a ONE-channel map into 128 channels (depth_models.py:19,36) and its
// weight + bias gradient.  With C_in = 1 the contraction is only 49 deep, so the MFMA formulation spends its time in the
// scalar gather loader (121 us forward / 158 us weight gradient at batch 64); here it is plain fp32 vector arithmetic:
//   * a wave owns one output pixel at a time, lane j owns channels j and j+64 as a float2 -> v_pk_fma_f32 (2 FMA / lane /
//     instruction, the full-rate packed fp32 path of CDNA3/4);
//   * workgroups are persistent over output rows (b, oh): the 7 input rows of an output row are copied into LDS by
//     LDS-DMA two rows ahead (three stages); a wave handles 4 adjacent pixels at a time, whose 16 taps per image row are
//     wave-uniform LDS broadcast reads (28 FMAs per 16 reads);
//   * forward keeps the 49 weights of its two channels in registers (fetched once per workgroup through LDS, coalesced),
//     the weight gradient keeps 49 float2 accumulators.
// Both are bound by the 128-channel fp32 activation stream (write of y / read of dy: 179 MB at batch 64).
// Exact fp32 FMA arithmetic; summation order per output differs from the MFMA kernel only in the usual fp32 round-off.
//
// Reproducibility (round 2).  In their first form both kernels were bit-reproducible alone on the chip but not next to another
// stream's kernels (a concurrent bf16x3 ResNet forward): one pixel in ~10^5 came out with channels 48..63 - the x half of lanes
// 48..63 - off by up to 0.1, in registers (the BatchNorm partial sums saw it too); 59 of 59 repetitions differed
// (scripts/diag_conv1_race.py with codes 70 75).  Cause (scripts/diag_pk_fp32_opsel.py, a register-only reproducer): on gfx950
// a packed fp32 instruction whose LOW result takes the HIGH half of src1 - `v_pk_fma_f32 ... op_sel:[0,1,0]`, which hipcc chose
// for 69 of the 734 tap broadcasts because the tap sat in the upper register of a ds_read2_b32 pair - gives wrong results in
// ~1e-5 of the lanes when the wave shares its SIMD with other kernels' waves; every other operand-select form is right.
// Fix: each tap is pinned in a register of its own (asm "+v") before the row's FMAs, so a broadcast always reads the LOW
// register of the pair the instruction names (op_sel_hi:[1,0,1], the safe form); build.py audits the device assembly of every
// build for the bad forms, tests/test_encoders_gpu.py::test_layer1_kernels_reproducible_next_to_lds_heavy_kernels repeats the
// failing scenario.  Cost: a few microseconds per launch; still 0.03 + 0.08 ms per step better than the generic gather kernels.
#include "conv.h"
#include "nn_kernels.h"

namespace dic {

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kC1MaxW = 640;         // widest input row staged in LDS (3 stages x 7 rows x W floats must fit 64 KB)
constexpr int kTapPad = 16;          // floats readable behind a stage (ragged last pixel group)

__host__ __device__ inline int c1_stage_floats(int ks, int W) { return ((ks * W + 63) & ~63) + kTapPad; }

// Asynchronous copy of the KS contiguous image rows under output row `row` into an LDS stage (LDS-DMA, one dword per
// lane, 64 consecutive floats per wave instruction; lanes past the end re-read the last float into the stage's slack).
template <int KS, int S>
__device__ __forceinline__ int c1_stage_rows(const float* __restrict__ x, int H, int W, int OH, int row, float* stage,
                                              int wv, int lane) {
  const int b = row / OH, oh = row - b * OH;
  const float* src = x + ((long long)b * H + (long long)oh * S) * W;
  const int n = KS * W;
  int issued = 0;
  for (int c = wv * 64; c < n; c += 256, ++issued) {
    const float* g = src + min(c + lane, n - 1);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(stage + c), 4, 0, 0);
  }
  return issued;
}

// s_waitcnt vmcnt(n) for a wave-uniform runtime n (the instruction only takes an immediate): rounds n DOWN to a multiple
// of 8, which is always safe (waits for more).  vmcnt retires in order, so "at most n outstanding" = "everything older
// than the n newest vector-memory operations has completed".
__device__ __forceinline__ void c1_wait_vm_at_most(int n) {
  n = __builtin_amdgcn_readfirstlane(n);
  if (n >= 56) asm volatile("s_waitcnt vmcnt(56)" ::: "memory");
  else if (n >= 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
  else if (n >= 40) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
  else if (n >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
  else if (n >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// y[row][ow][c] = bias[c] + sum_{kh,kw} w[c][kh][kw] * x[b][oh*S+kh][ow*S+kw];  workgroups stride over the output rows
// row = (b, oh); the input rows of output rows k+1 and k+2 are in flight (three LDS stages) while row k is computed.
// partial (nullable): per-WORKGROUP BatchNorm sums [gridDim.x][2][128] of the stored values.
template <int KS, int S>
__global__ void __launch_bounds__(256, 2) conv1_fwd_kernel(const float* __restrict__ x, int H, int W, int OH, int OW,
                                                           int rows_total, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           float* __restrict__ partial) {
  constexpr int KK = KS * KS, PG = 4;
  extern __shared__ float smem[];      // max(128*KK weights, 3 stages), then [4][2][128] statistics
  const int tid = threadIdx.x, j = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int SL = c1_stage_floats(KS, W);
  float(*red)[2][128] = reinterpret_cast<float(*)[2][128]>(smem + max(128 * KK, 3 * SL));
  for (int i = tid; i < 128 * KK; i += 256) smem[i] = w[i];
  __syncthreads();
  f2 wr[KK];
#pragma unroll
  for (int k = 0; k < KK; ++k) { wr[k].x = smem[j * KK + k]; wr[k].y = smem[(j + 64) * KK + k]; }   // stride 49: conflict-free
  f2 bs;
  bs.x = bias ? bias[j] : 0.f; bs.y = bias ? bias[j + 64] : 0.f;
  f2 s = {0.f, 0.f}, s2 = {0.f, 0.f};
  __syncthreads();                                                          // weights consumed: stages may be filled
  const int G = gridDim.x;
  // Row pipeline (vmcnt retires in order, and a store only retires when it is acknowledged - microseconds under load):
  // the copy of row k+2 is issued at the END of row k, after that row's stores; the top of row k+1 then waits until
  // at most [stores of row k] + [copy k+2] operations are outstanding, i.e. exactly until copy k+1 (issued a whole row
  // earlier) has landed, without ever waiting for a recent store.
  if ((int)blockIdx.x < rows_total) c1_stage_rows<KS, S>(x, H, W, OH, blockIdx.x, smem, wv, j);
  int newer = 0;                                          // vector-memory operations issued after the copy we wait for
  if ((int)blockIdx.x + G < rows_total) newer = c1_stage_rows<KS, S>(x, H, W, OH, blockIdx.x + G, smem + SL, wv, j);
  int it = 0;
  for (int row = blockIdx.x; row < rows_total; row += G, ++it) {
    c1_wait_vm_at_most(newer);
    __syncthreads();                                      // row `row` landed for every wave; stage (it + 2) % 3 is free
    int stores = 0;
    const float* cur = smem + (it % 3) * SL;
    // a wave takes PG adjacent pixels at a time: their taps overlap (stride S < KS), so one row of (PG-1)*S+KS LDS
    // broadcast reads serves PG*KS FMAs.  The last group of a row is ragged: its surplus taps come from the next staged
    // row / the slack behind the stage and only reach accumulators of pixels that are never stored
#pragma unroll 1
    for (int og = wv * PG; og < OW; og += 4 * PG) {
      f2 acc[PG];
#pragma unroll
      for (int q = 0; q < PG; ++q) acc[q] = bs;
      const float* p = cur + og * S;
      constexpr int NT = (PG - 1) * S + KS;
#pragma unroll
      for (int kh = 0; kh < KS; ++kh) {
        float t[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) t[u] = p[kh * W + u];
        // every tap in a register of its own: broadcasts then use the safe operand-select form (note at the top of the file)
#pragma unroll
        for (int u = 0; u < NT; ++u) asm volatile("" : "+v"(t[u]));
#pragma unroll
        for (int q = 0; q < PG; ++q)
#pragma unroll
          for (int kw = 0; kw < KS; ++kw) {
            const f2 vv = {t[q * S + kw], t[q * S + kw]};
            acc[q] = __builtin_elementwise_fma(wr[kh * KS + kw], vv, acc[q]);
          }
      }
      // all PG accumulators are consumed unconditionally (masked statistics) so the compiler keeps the FMAs in this
      // straight-line block instead of sinking them into the per-pixel store branches
#pragma unroll
      for (int q = 0; q < PG; ++q) {
        const float m = og + q < OW ? 1.f : 0.f;
        const f2 mm = {m, m};
        s = __builtin_elementwise_fma(acc[q], mm, s);
        s2 = __builtin_elementwise_fma(acc[q] * acc[q], mm, s2);
        if (og + q < OW) {
          float* o = y + ((long long)row * OW + og + q) * 128;
          o[j] = acc[q].x; o[j + 64] = acc[q].y;
          stores += 2;
        }
      }
    }
    newer = stores;
    if (row + 2 * G < rows_total)
      newer += c1_stage_rows<KS, S>(x, H, W, OH, row + 2 * G, smem + ((it + 2) % 3) * SL, wv, j);
  }
  if (partial) {
    red[wv][0][j] = s.x; red[wv][0][j + 64] = s.y; red[wv][1][j] = s2.x; red[wv][1][j + 64] = s2.y;
    __syncthreads();
    const int c = tid & 127, q = tid >> 7;
    partial[((long long)blockIdx.x * 2 + q) * 128 + c] = (red[0][q][c] + red[1][q][c]) + (red[2][q][c] + red[3][q][c]);
  }
}

// Per-workgroup partial of dW[c][k] = sum_pixels dy[pixel][c] * tap_k(pixel) and db[c] = sum_pixels dy[pixel][c]:
// ws[blockIdx.x][0 .. 128*KS*KS) = dW partial (OIHW order), [128*KS*KS .. +128) = db partial.  Row pipeline: the image
// rows of the NEXT output row are loaded into NPRE registers per thread before this row's gradient loads are issued (so
// no gradient load ever queues behind them: vmcnt retires in order) and written to the other LDS stage after the row;
// the four waves' accumulators are combined through LDS in a fixed order.
template <int KS, int S, int NPRE>
__global__ void __launch_bounds__(256, 3) conv1_wgrad_kernel(const float* __restrict__ x, int H, int W, int OH, int OW,
                                                             int rows_total, const float* __restrict__ dy,
                                                             float* __restrict__ ws) {
  constexpr int KK = KS * KS, ROWLEN = 128 * KK + 128, PG = 4;
  extern __shared__ float smem[];      // max(ROWLEN reduction buffer, 2 stages)
  const int tid = threadIdx.x, j = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int SL = c1_stage_floats(KS, W), n = KS * W, n64 = (n + 63) & ~63;
  f2 acc[KK];
#pragma unroll
  for (int k = 0; k < KK; ++k) acc[k] = f2{0.f, 0.f};
  f2 gsum = {0.f, 0.f};
  // surplus taps of a ragged last group meet zero gradients: the slack behind each stage must hold finite values
  if (tid < 2 * kTapPad) smem[(tid / kTapPad) * SL + SL - kTapPad + tid % kTapPad] = 0.f;
  const int G = gridDim.x;
  float pre[NPRE];
  auto load_rows = [&](int row) {
    const int b = row / OH, oh = row - b * OH;
    const float* src = x + ((long long)b * H + (long long)oh * S) * W;
#pragma unroll
    for (int u = 0; u < NPRE; ++u) pre[u] = src[min(tid + 256 * u, n - 1)];
  };
  auto store_rows = [&](float* stage) {
#pragma unroll
    for (int u = 0; u < NPRE; ++u)
      if (tid + 256 * u < n64) stage[tid + 256 * u] = pre[u];
  };
  if ((int)blockIdx.x < rows_total) { load_rows(blockIdx.x); store_rows(smem); }
  int it = 0;
  for (int row = blockIdx.x; row < rows_total; row += G, ++it) {
    __syncthreads();                                  // stage it & 1 is complete; everyone left stage (it + 1) & 1
    const bool has_next = row + G < rows_total;
    if (has_next) load_rows(row + G);
    const float* cur = smem + (it & 1) * SL;
    const float* drow = dy + (long long)row * OW * 128;
#pragma unroll 1
    for (int og = wv * PG; og < OW; og += 4 * PG) {
      f2 g[PG];
#pragma unroll
      for (int q = 0; q < PG; ++q) {             // pixels past the row end contribute zero (clamped load, then select)
        const int ow = min(og + q, OW - 1);
        const float gx = drow[(long long)ow * 128 + j], gy = drow[(long long)ow * 128 + j + 64];
        g[q].x = og + q < OW ? gx : 0.f; g[q].y = og + q < OW ? gy : 0.f;
        gsum += g[q];
      }
      const float* p = cur + og * S;
      constexpr int NT = (PG - 1) * S + KS;
      float t[NT], tn[NT];
#pragma unroll
      for (int u = 0; u < NT; ++u) t[u] = p[u];
#pragma unroll
      for (int u = 0; u < NT; ++u) asm volatile("" : "+v"(t[u]));
#pragma unroll
      for (int kh = 0; kh < KS; ++kh) {                 // row kh+1's taps are in flight during row kh's FMAs
        if (kh + 1 < KS) {
#pragma unroll
          for (int u = 0; u < NT; ++u) tn[u] = p[(kh + 1) * W + u];
#pragma unroll
          for (int u = 0; u < NT; ++u) asm volatile("" : "+v"(tn[u]));      // (same)
        }
#pragma unroll
        for (int q = 0; q < PG; ++q)
#pragma unroll
          for (int kw = 0; kw < KS; ++kw) {
            const f2 vv = {t[q * S + kw], t[q * S + kw]};
            acc[kh * KS + kw] = __builtin_elementwise_fma(g[q], vv, acc[kh * KS + kw]);
          }
        __builtin_amdgcn_sched_barrier(0);              // keeps the 7 rows of reads from being hoisted together
#pragma unroll
        for (int u = 0; u < NT; ++u) t[u] = tn[u];
      }
    }
    if (has_next) store_rows(smem + ((it + 1) & 1) * SL);
  }
  __syncthreads();
  float* red = smem;
  for (int r = 0; r < 4; ++r) {
    if (wv == r) {
      if (r == 0) {
#pragma unroll
        for (int k = 0; k < KK; ++k) { red[j * KK + k] = acc[k].x; red[(j + 64) * KK + k] = acc[k].y; }
        red[128 * KK + j] = gsum.x; red[128 * KK + j + 64] = gsum.y;
      } else {
#pragma unroll
        for (int k = 0; k < KK; ++k) { red[j * KK + k] += acc[k].x; red[(j + 64) * KK + k] += acc[k].y; }
        red[128 * KK + j] += gsum.x; red[128 * KK + j + 64] += gsum.y;
      }
    }
    __syncthreads();
  }
  float* out = ws + (long long)blockIdx.x * ROWLEN;
  for (int i = tid; i < ROWLEN; i += 256) out[i] = red[i];
}

static int g_c1_blocks = 512;     // persistent workgroups (<= 512 keeps the BatchNorm finalize a single launch)
static int g_c1_on = 1;            // debug codes 131..134 = on with 256..1024 workgroups (default 512), 130 = generic gather kernels instead
void conv1_depth_debug_blocks(int n) { if (n <= 0) { g_c1_on = 0; return; } g_c1_on = 1; g_c1_blocks = n > 1024 ? 1024 : n; }
bool conv1_depth_enabled() { return g_c1_on != 0; }

static bool conv1_shape_ok(const ConvDesc& d) {
  return d.C == 1 && d.CO == 128 && d.KH == 7 && d.KW == 7 && d.stride == 3 && d.pad == 0 && d.OH() >= 1 && d.OW() >= 1 &&
         d.W <= kC1MaxW;
}

bool conv1_depth_supported(const ConvDesc& d) { return g_c1_on != 0 && conv1_shape_ok(d); }

int conv1_depth_fwd_blocks(const ConvDesc& d) { return std::min(d.B * d.OH(), g_c1_blocks); }

int conv1_depth_fwd(const float* x, const ConvDesc& d, const float* w, const float* bias, float* y, float* bn_partial,
                    int* partial_rows, hipStream_t st) {
  DIC_REQUIRE(conv1_shape_ok(d), "conv1_depth_fwd: expects a 1->128 channel 7x7 stride-3 unpadded convolution");
  const int rows = d.B * d.OH(), nb = conv1_depth_fwd_blocks(d);
  const size_t lds = (size_t)(std::max(128 * 49, 3 * c1_stage_floats(7, d.W)) + 4 * 2 * 128) * sizeof(float);
  hipLaunchKernelGGL((conv1_fwd_kernel<7, 3>), dim3(nb), dim3(256), lds, st, x, d.H, d.W, d.OH(), d.OW(), rows, w, bias, y,
                     bn_partial);
  DIC_LAUNCH_CHECK();
  if (partial_rows) *partial_rows = nb;
  return DIC_OK;
}

int conv1_depth_wgrad_blocks(const ConvDesc& d) { return std::min(d.B * d.OH(), g_c1_blocks); }
size_t conv1_depth_wgrad_ws_floats(const ConvDesc& d) { return (size_t)std::min(d.B * d.OH(), 1024) * (128 * 49 + 128); }

int conv1_depth_wgrad(const float* x, const ConvDesc& d, const float* dy, float* dw, float* dbias, float* ws, float* cs_ws,
                      hipStream_t st) {
  DIC_REQUIRE(conv1_shape_ok(d), "conv1_depth_wgrad: expects a 1->128 channel 7x7 stride-3 unpadded convolution");
  const int nb = conv1_depth_wgrad_blocks(d);
  constexpr int ROWLEN = 128 * 49 + 128;
  const size_t lds = (size_t)std::max(ROWLEN, 2 * c1_stage_floats(7, d.W)) * sizeof(float);
  const int rows = d.B * d.OH();
  if (d.W <= 256)
    hipLaunchKernelGGL((conv1_wgrad_kernel<7, 3, 7>), dim3(nb), dim3(256), lds, st, x, d.H, d.W, d.OH(), d.OW(), rows, dy, ws);
  else if (d.W <= 512)
    hipLaunchKernelGGL((conv1_wgrad_kernel<7, 3, 14>), dim3(nb), dim3(256), lds, st, x, d.H, d.W, d.OH(), d.OW(), rows, dy, ws);
  else
    hipLaunchKernelGGL((conv1_wgrad_kernel<7, 3, 18>), dim3(nb), dim3(256), lds, st, x, d.H, d.W, d.OH(), d.OW(), rows, dy, ws);
  DIC_LAUNCH_CHECK();
  DIC_TRY(colsum_rows(ws, ROWLEN, nb, 128 * 49, dw, cs_ws, st));
  if (dbias) DIC_TRY(colsum_rows(ws + 128 * 49, ROWLEN, nb, 128, dbias, cs_ws, st));
  return DIC_OK;
}

}  // namespace dic
