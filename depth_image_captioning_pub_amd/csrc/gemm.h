// Generic exact-fp32 MFMA contraction for gfx950:  C[m,n] (+)= sum_k A(m,k) * B(n,k)
//
// One kernel template covers every contraction of the hot path (linear layers, implicit-GEMM
// convolution forward / data-gradient / weight-gradient) by swapping the operand *loaders*:
//   ROWK         A(i,k) = p[i*ld + k]              (K contiguous: activations x, weights [out][in])
//   COLK         A(i,k) = p[k*ld + i]              (K is the slow index: "transposed" operands)
//   IM2COL       A(m,k) = x[img, oh*s-p+kh, ow*s-p+kw, c]  NHWC, k=(kh,kw,c), C % 32 == 0
//   GATHER       same, any C / NCHW input, scalar loads (7x7 stem convs with C_in = 1 or 3)
//   IM2COL_COLK  B(j,k=m) = x[img(m), oh+kh.., c]  j=(kh,kw,c): weight-gradient B operand, C % 4 == 0
//   GATHER_COLK  same, scalar (C_in = 1)
// Arithmetic: v_mfma_f32_32x32x2_f32 (exact f32 in / f32 accumulate, 256 FLOP/clk/CU), which is
// what the parity bar (loss within 1e-4, argmax bit-exact vs the fp32 CPU path) needs.
// Tile: BMxBNx32, 4 waves (2x2), LDS k-major [32][BM+pad] so every fragment read is a
// conflict-free ds_read_b32 (lanes 0-31 = 32 consecutive rows, lanes 32-63 = next k);
// register-staged double buffering, one barrier per K tile.
#pragma once
#include "common.h"

namespace dic {

// Element offset of (row r, column k) inside one bf16 plane of the bf16x3 operand format; kblocks = K / 32.
//   paired = 0: plain row-major [rows][K].
//   paired = 1: rows are stored in pairs, interleaved per 32-element K block: line (r >> 1, k >> 5) holds
//               [row 2q: 32 elements | row 2q+1: 32 elements] = 128 B, so the 16 rows x 64 B one LDS-DMA instruction
//               moves are 8 whole cache lines rather than 16 half lines (+8..29 % kernel throughput, measured).
//               A plane of `rows` rows occupies ((rows + 1) & ~1) * K elements.
__host__ __device__ __forceinline__ long long plane_offset(long long r, long long k, long long kblocks, int paired) {
  if (!paired) return r * kblocks * 32 + k;
  return (((r >> 1) * kblocks + (k >> 5)) << 6) + ((r & 1) << 5) + (k & 31);
}

enum : int { OPK_ROWK = 0, OPK_COLK = 1, OPK_IM2COL = 2, OPK_GATHER = 3, OPK_IM2COL_COLK = 4, OPK_GATHER_COLK = 5 };
enum : int { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2, ACT_GELU = 3 };   // GELU: exact (erf) form, nn.GELU()

struct ConvGeom {
  int H, W, C;      // input height / width / channels
  int OH, OW;       // output height / width
  int KH, KW, stride, pad;
  int nchw;         // GATHER only: input stored NCHW instead of NHWC
};

struct GemmOperand {
  const float* p;
  long long ld;
  int kind;
  int vec;          // 16-byte vector loads are legal (alignment + ld % 4 == 0)
  ConvGeom g;
};

struct GemmEpilogue {
  float* C;             // output, row stride ldc
  long long ldc;
  const float* bias;    // per output column, or null
  int act;
  int accumulate;       // C += result
  const int* row_map;   // optional: output row index = row_map[m]
  float* stats;         // optional BN partials: [mtiles][2][N] (sum, sum of squares of the stored value)
  float* C2;            // optional column split: columns >= nsplit go to C2[m*ldc2 + (n-nsplit)]
  long long ldc2;
  int nsplit;
  float alpha;          // result scale (applied before bias)
  // two more factors of the result scale that live in DEVICE memory (nullable): the inverse operand scales of f16x2 planes whose scale
  // is chosen on the device (trained weights, gradients: the depth encoder).  Effective scale = alpha * *alpha_dev[0] * *alpha_dev[1];
  // all three are powers of two, so the product is exact.  Read once per workgroup (ep_alpha, gemm_epilogue.h).
  const float* alpha_dev[2];
};

struct GemmParams {
  int M, N, K;
  GemmOperand A, B;
  GemmEpilogue ep;
  int mtiles, ntiles;
  int splitk;           // >1: raw partials go to ws[z][M][N]; splitk_reduce applies the epilogue
  int ktiles_per_split;
  float* ws;
  // remainder ('tail') tiles: blocks >= tail_first_block each compute 1/tail_split of the K range of one of
  // the last tiles and leave raw partials in tail_ws; tail_fixup_kernel finishes those tiles (gemm.hip)
  int tail_first_block, tail_first_tile, tail_split;
  float* tail_ws;
  int ablate;           // debug/benchmark only: 1 no global traffic in the loop, 2 also no LDS reads, 3 LDS stores but no global loads
  int raw_partials;     // split-K: leave the [splitk][M][N] partial slabs in ws, skip the reduce launch
  // tail fix-up of the 128x128 persistent kernels (gemm_bf3.hip): tail index = 4 * (remainder tile - tail128_first) + quadrant,
  // the 64x64 tile it finishes = quadrant (q >> 1, q & 1) of 128x128 tile tail128_first + index / 4 on a grid tail128_ntiles wide
  int tail128_first, tail128_ntiles;
};

// host side (gemm.hip)
size_t gemm_splitk_ws_bytes(int M, int N, int splitk);
constexpr size_t kGemmTailWsBytes = (size_t)256 * 64 * 64 * sizeof(float);   // tail_ws capacity needed by gemm_launch
int gemm_pick_tile(int M, int N);               // 128 or 64
int gemm_launch(GemmParams p, hipStream_t st, int force_tile = 0);
constexpr int kGemmGroupMax = 6;
// several independent K-major x K-major contractions (64x64 tiles) in one launch; ps[i] is completed in place (gemm.hip)
int gemm_launch_group_colk(GemmParams* ps, int n, hipStream_t st);
GemmOperand op_rowk(const float* p, long long ld);
GemmOperand op_colk(const float* p, long long ld);
GemmOperand op_im2col(const float* x, const ConvGeom& g);
GemmOperand op_gather(const float* x, const ConvGeom& g);
GemmOperand op_im2col_colk(const float* x, const ConvGeom& g);
GemmOperand op_gather_colk(const float* x, const ConvGeom& g);
void gemm_force_v1(int on);
int gemm_bf3_set_persist_grid(int workgroups);
int gemm_bf3_force_tile(int code);      // 0 = accepted, -1 = unknown code in this build
int gemm_launch_tail_fixup(const GemmParams& p, int tail_tiles, hipStream_t st);

// Train-mode BatchNorm finalize fused into the tail fix-up launch (saves one dependent dispatch per tail-split
// convolution): extra workgroups reduce the tile partials of 32 channels each; for the remainder tiles they recompute
// the statistics from the K slices themselves, so they do not depend on the fix-up workgroups of the same launch.
struct BnFuseArgs {
  const float *gamma, *beta;
  float *running_mean, *running_var;      // nullable
  float *scale, *shift, *mean, *invstd;   // outputs (BnBuf)
  double count;                           // rows of the activation (B*OH*OW)
  float eps, momentum;
  unsigned* status = nullptr;             // f16x2 overflow guard word (common.h): raised on non-finite statistics
};
// requires p.ep.stats, tail tiles aligned to whole tile rows (tail_first_tile % ntiles == 0), no bias / activation.
bool gemm_tail_fixup_bn_eligible(const GemmParams& p, int tail_tiles);
int gemm_launch_tail_fixup_bn(const GemmParams& p, int tail_tiles, const BnFuseArgs& bn, hipStream_t st);
void gemm_profile_mark_begin(hipStream_t st, double flops, int key, double bytes = 0.0 /* algorithmic HBM bytes of the launch */);
void gemm_profile_mark_end(hipStream_t st);
int gemm_profile_begin();
int gemm_profile_end(int max_entries, int* keys, double* total_ms, double* total_flops, long long* launches, int* n_out,
                     double* total_bytes = nullptr);
GemmEpilogue ep_store(float* C, long long ldc, const float* bias = nullptr, int act = ACT_NONE);

// convenience: C = A(MxK, rowk) * B(NxK, rowk)^T etc. with automatic split-K for skinny shapes
struct GemmCall {
  GemmParams p{};
  GemmCall(int M, int N, int K, GemmOperand A, GemmOperand B, GemmEpilogue ep) {
    p.M = M; p.N = N; p.K = K; p.A = A; p.B = B; p.ep = ep; p.splitk = 1; p.ws = nullptr;
  }
};

}  // namespace dic
