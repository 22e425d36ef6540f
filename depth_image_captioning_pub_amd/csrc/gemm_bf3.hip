// fp32-accurate contraction on the bf16 matrix cores ("bf16x3"): every fp32 operand is stored as three bf16
// planes hi + mid + lo (an EXACT decomposition: 3 x 8 significand bits = fp32's 24), and a product a*b is formed
// as the six bf16 x bf16 products  ah*bh + ah*bm + am*bh + am*bm + ah*bl + al*bh,  each exact in fp32 and
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The dropped terms (am*bl, al*bm, al*bl) are <= 2^-24 |ab|,
// i.e. one fp32 unit round-off, so the result carries fp32-level error (checked against fp64 in the tests) at
// 6 MFMAs x 32 cycles per K=16 versus 8 x 64 cycles for the exact-fp32 MFMA: 2.67x the matrix-core rate.
//
// Kernel structure = gemm_dma_kernel (gemm.hip): 64x64 tile, 4 waves, LDS-DMA staging with source-side swizzle,
// 2-stage ring, inline-asm fragment reads with counted lgkmcnt, DMA issue in the MFMA shadow.
#include "conv.h"
#include "gemm_epilogue.h"
#include <algorithm>
#include <type_traits>
#include <cstdlib>

namespace dic {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// Operand formats of this file's kernels (template parameter FMT):
//   0  "bf16x3": three bf16 planes, six products (above) - exact decomposition, fp32-level result
//   1  "f16x2":  two fp16 planes h1 + h2 of s*x (s a power of two that puts the operand's values high in the fp16 range;
//      h1 = rn(s*x), h2 = rn(s*x - h1): |s*x - h1 - h2| <= 2^-22 |s*x|, subnormal h2 are honoured by the matrix cores - checked,
//      scripts/micro/mfma_f16_subnormal.hip) and the three products h1*h1' + h1*h2' + h2*h1' (dropped: h2*h2' <= 2^-22 |ab|):
//      a few fp32 round-offs per product instead of one, at half the matrix-core work and two thirds of the operand bytes.
//      The epilogue multiplies the accumulators by ep.alpha = 1 / (s_a * s_b) (exact).
template <int FMT> struct Bf3Fmt { static constexpr int NPL = FMT == 1 ? 2 : 3; };
constexpr int BK3 = 32;          // K tile in elements (64 B per plane row)
#ifndef DIC_WS_A_AUX
#define DIC_WS_A_AUX 0    // cache policy of the streamed A operand in the warp-specialised kernel (2 = nt measured within +-3 %: left at default)
#endif

__device__ __attribute__((aligned(256))) const unsigned short g_zero_line16[128] = {0};

__global__ void __launch_bounds__(256) split_bf16x3_kernel(const float* __restrict__ x, long long n,
                                                            unsigned short* __restrict__ hi,
                                                            unsigned short* __restrict__ mid,
                                                            unsigned short* __restrict__ lo) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    unsigned short h, m, l;
    split3_bf16(x[i], h, m, l);
    hi[i] = h; mid[i] = m; lo[i] = l;
  }
}

// Same split, written in the row-pair interleaved layout (plane_offset, gemm.h): 16 threads fill one 128-B line.
__global__ void __launch_bounds__(256) split_bf16x3_paired_kernel(const float* __restrict__ x, long long rows, int K,
                                                                   unsigned short* __restrict__ hi,
                                                                   unsigned short* __restrict__ mid,
                                                                   unsigned short* __restrict__ lo) {
  const long long n4 = ((rows + 1) >> 1) * (K / 2);          // float4 slots of the padded plane
  const int kb = K / 32;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const long long line = i >> 4;
    const int j = (int)(i & 15);
    const long long r = (line / kb) * 2 + (j >> 3);
    const int k = (int)(line % kb) * 32 + (j & 7) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows) v = *reinterpret_cast<const float4*>(x + r * K + k);
    unsigned short h[4], m[4], l[4];
    split3_bf16(v.x, h[0], m[0], l[0]); split3_bf16(v.y, h[1], m[1], l[1]);
    split3_bf16(v.z, h[2], m[2], l[2]); split3_bf16(v.w, h[3], m[3], l[3]);
    reinterpret_cast<uint2*>(hi)[i] = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
    reinterpret_cast<uint2*>(mid)[i] = make_uint2((unsigned)m[0] | ((unsigned)m[1] << 16), (unsigned)m[2] | ((unsigned)m[3] << 16));
    reinterpret_cast<uint2*>(lo)[i] = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
  }
}

// f16x2 format (top of the file), same row-pair interleaved layout: two planes of scale * x
__global__ void __launch_bounds__(256) split_f16x2_paired_kernel(const float* __restrict__ x, long long rows, int K, float scale,
                                                                  unsigned short* __restrict__ h1, unsigned short* __restrict__ h2,
                                                                  unsigned* __restrict__ status) {
  const long long n4 = ((rows + 1) >> 1) * (K / 2);
  const int kb = K / 32;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const long long line = i >> 4;
    const int j = (int)(i & 15);
    const long long r = (line / kb) * 2 + (j >> 3);
    const int k = (int)(line % kb) * 32 + (j & 7) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows) v = *reinterpret_cast<const float4*>(x + r * K + k);
    unsigned short a[4], b[4];
    if (f16x2_out_of_range(v.x, scale) | f16x2_out_of_range(v.y, scale) | f16x2_out_of_range(v.z, scale) | f16x2_out_of_range(v.w, scale))
      f16x2_raise(status, 16u);           // overflow guard (common.h)
    split2_f16(v.x, scale, a[0], b[0]); split2_f16(v.y, scale, a[1], b[1]);
    split2_f16(v.z, scale, a[2], b[2]); split2_f16(v.w, scale, a[3], b[3]);
    reinterpret_cast<uint2*>(h1)[i] = make_uint2((unsigned)a[0] | ((unsigned)a[1] << 16), (unsigned)a[2] | ((unsigned)a[3] << 16));
    reinterpret_cast<uint2*>(h2)[i] = make_uint2((unsigned)b[0] | ((unsigned)b[1] << 16), (unsigned)b[2] | ((unsigned)b[3] << 16));
  }
}

struct Bf3Operand {
  const unsigned short* p[3];   // hi, mid, lo planes, element (i,k) at p[.][i*ld + k]  (or NHWC image for im2col)
  long long ld;
  int kind;                     // OPK_ROWK or OPK_IM2COL
  int paired;                   // row-pair interleaved storage (see plane_offset): DMA fetches whole 128-B lines
  ConvGeom g;
};

struct Bf3Params {
  int M, N, K;
  Bf3Operand A, B;
  GemmEpilogue ep;
  int mtiles, ntiles;
  int splitk;                   // always 1 (field layout shared with gemm_epilogue)
  float* ws;
  int tail_first_block, tail_first_tile, tail_split;   // remainder-tile K split, as in gemm.hip (64x64 tile only)
  float* tail_ws;
  // A operand computed on the fly (OPK_ROWK_BN, warp-specialised kernel only): A(m,k) = act(a_raw[m][k] * a_scale[k] + a_shift[k]
  // (+ a_res[m][k])) - the BatchNorm-apply / residual / ReLU pass that would otherwise write it as planes; a_out (nullable)
  // receives the fp32 values once (tiles with tn == 0): the block output that is the next block's identity
  const float *a_raw, *a_scale, *a_shift, *a_res;
  // f16x2 form of the 1x1 kernel only (nullable): the residual is itself a raw convolution output with a BatchNorm of its own (the
  // downsample branch of a stage's first block): residual = a_res * a_res_scale[k] + a_res_shift[k], one fused multiply-add rounded to
  // fp32 - what the in-place bn_apply pass over that branch used to leave in memory
  const float *a_res_scale, *a_res_shift;
  float* a_out;
  long long a_ld;
  int a_relu;
  int fmt;                      // operand format of A and B: 0 = bf16x3, 1 = f16x2 (ep.alpha then carries 1 / (scale_a * scale_b))
  unsigned* status;             // f16x2 overflow guard word (common.h), nullable: raised by the producer waves of the on-the-fly operand
  int few_remap;                // few-tiles launches: XCD-aware (tile, slice) assignment (few_tiles_remap)
};
constexpr int OPK_ROWK_BN = 6;     // (A-operand kind of the kernel template; never stored in Bf3Operand::kind)
constexpr int kBnTabMax = 2048;    // channels of the on-the-fly operand (its scale / shift table lives in LDS)

// One operand's DMA bookkeeping: a wave-instruction fills 16 rows x 64 B of one plane image; wave w takes the row
// groups w, w+4, ... of each of the three planes.  Address arithmetic is done once per output tile.
template <int KIND, int BR, int NPL = 3>
struct Bf3Loader {
  static constexpr int NI = BR / 64;
  const unsigned short* p[3];
  int K, C, KW, W, paired;
  int strip = 0;              // strip mode: elements per padded image row (0 = ordinary im2col)
  int pix0[NI];               // paired im2col: linear input pixel of tap (0,0) (may be negative: masked by tapmask)
  long long lane_off[NI];     // elements
  unsigned tapmask[NI];
  bool valid[NI];
  int kchunk[NI];

  __device__ __forceinline__ void init(const Bf3Operand& op, int r0, int R, int K_) {
    p[0] = op.p[0]; p[1] = op.p[1]; p[2] = op.p[2];
    K = K_; C = op.g.C; KW = op.g.KW; W = op.g.W; paired = op.paired;
    const int lane = threadIdx.x & 63, w = (threadIdx.x >> 6) & 3;      // (& 3: producer waves 4..7 of the warp-specialised kernels)
#pragma unroll
    for (int n = 0; n < NI; ++n) {
      const int row = (n * 4 + w) * 16 + (lane >> 2);
      const int gr = r0 + row;
      valid[n] = gr < R;
      const int chunk = (lane & 3) ^ ((row >> 2) & 3);
      kchunk[n] = chunk * 8;
      tapmask[n] = 0u;
      if constexpr (KIND == OPK_ROWK) {
        lane_off[n] = op.paired ? plane_offset(gr, chunk * 8, op.ld / 32, 1) : (long long)gr * op.ld + chunk * 8;
      } else {
        const ConvGeom& g = op.g;
        const int ohw = g.OH * g.OW;
        const int img = gr / ohw, rem = gr - img * ohw;
        const int oh = rem / g.OW, ow = rem - oh * g.OW;
        const int ih0 = oh * g.stride - g.pad, iw0 = ow * g.stride - g.pad;
        lane_off[n] = ((long long)img * g.H + ih0) * g.W * g.C + (long long)iw0 * g.C + chunk * 8;
        pix0[n] = (img * g.H + ih0) * g.W + iw0;
        if (g.nchw == 2) {      // strip mode (7x7 stem on a zero-padded NHWC4 image): K tile kh = the 32 contiguous
          strip = g.W * g.C;    // elements (8 pixels x 4 channels) of input row ih0 + kh starting at pixel iw0
          tapmask[n] = valid[n] ? 0xffffffffu : 0u;
          continue;
        }
        unsigned m = 0u;
        for (int kh = 0; kh < g.KH; ++kh)
          for (int kw = 0; kw < g.KW; ++kw)
            if ((unsigned)(ih0 + kh) < (unsigned)g.H && (unsigned)(iw0 + kw) < (unsigned)g.W) m |= 1u << (kh * g.KW + kw);
        tapmask[n] = valid[n] ? m : 0u;
      }
    }
  }

  // img: this operand's [3][BR][32] bf16 image of one stage
  template <int AUX = 0>      // AUX: cache-policy bits of the DMA (2 = nt, a streamed operand)
  __device__ __forceinline__ void issue(int k0, unsigned short* img) const {
    const int w = (threadIdx.x >> 6) & 3;
    int tap = 0, dpix = 0;
    long long uni = paired ? (long long)(k0 >> 5) * 64 : (long long)k0;
    if constexpr (KIND == OPK_IM2COL) {
      if (strip) {
        tap = 0;
        uni = (long long)(k0 >> 5) * strip;
      } else {
        tap = k0 / C;
        const int c0 = k0 - tap * C;
        const int kh = tap / KW, kw = tap - kh * KW;
        dpix = kh * W + kw;
        uni = paired ? (long long)(c0 >> 5) * 64 : ((long long)kh * W + kw) * C + c0;
      }
    }
#pragma unroll
    for (int n = 0; n < NI; ++n) {
      bool ok;
      if constexpr (KIND == OPK_ROWK) ok = valid[n] && (k0 + kchunk[n] < K);
      else ok = (tapmask[n] >> tap) & 1u;
      long long off = lane_off[n] + uni;
      if constexpr (KIND == OPK_IM2COL) {
        if (paired && !strip) {   // the tap moves the pixel, and with it the half of the 128-B pair line
          const int pix = pix0[n] + dpix;
          off = (long long)(pix >> 1) * (C * 2) + ((pix & 1) << 5) + kchunk[n] + uni;
        }
      }
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        const unsigned short* src = ok ? p[pl] + off : g_zero_line16;
        unsigned short* dst = img + pl * BR * BK3 + ((n * 4 + w) * 16) * BK3;       // wave-uniform 1-KiB block
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, AUX);
      }
    }
  }
};

// Row-major (OPK_ROWK) paired planes through buffer loads: per output tile one 32-bit byte offset per lane and row group; per K
// tile the instruction's scalar offset moves along K - no vector instructions at all in the K loop (the pointer form above
// spends ~7 per DMA on 64-bit address arithmetic, row clamping and the M0 value: ~80 per K tile and wave, issued on the SIMD the
// computing wave runs on).  Rows past the end of the operand get an offset beyond num_records: the load returns zeros (the
// scalar offset is not part of the range check on gfx9).  Needs K % 32 == 0 and planes below 2 GiB (launch_bf3 checks).
template <int BR, int NPL = 3>      // NPL: planes of the operand format
struct Bf3BufLoader {
  static constexpr int NI = BR / 64;
  // The buffer descriptor is REBUILT at every issue from scalars that pass through readfirstlane: inside the persistent kernels' work
  // loops hipcc cannot prove a descriptor kept in a struct across iterations wave-uniform (anything that met threadIdx-derived
  // control flow counts as divergent), parks it in vector registers and wraps every DMA in a "waterfall" loop - four readfirstlanes,
  // two 64-bit compares, saveexec, the load, a branch (guide T20).  The on-the-fly-operand kernel had 45 of them (r04: 141
  // v_readfirstlane in its listing), ~60 instructions per K tile on the producer waves, i.e. on the SIMDs the computing waves use.
  // A readfirstlane of a value that is already scalar costs nothing.
  const unsigned short* bp[3];
  int nbytes;
  unsigned voff[NI];
  int kstep;                  // bytes per K tile
  __device__ __forceinline__ void init(const Bf3Operand& op, int r0, int R, int) {
    const int kb = (int)(op.ld / 32);
    const long long bytes = op.paired ? (long long)((R + 1) / 2) * kb * 128 : (long long)R * op.ld * 2;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) bp[pl] = op.p[pl];
    nbytes = (int)bytes;
    kstep = op.paired ? 128 : 64;
    const int lane = threadIdx.x & 63, w = (threadIdx.x >> 6) & 3;
#pragma unroll
    for (int n = 0; n < NI; ++n) {
      const int row = (n * 4 + w) * 16 + (lane >> 2);
      const int gr = r0 + row;
      const int chunk = (lane & 3) ^ ((row >> 2) & 3);
      voff[n] = gr < R ? (unsigned)(plane_offset(gr, chunk * 8, kb, op.paired) * 2) : 0x80000000u;
    }
  }
  __device__ __forceinline__ void issue_plain(int k0, unsigned short* img) const {
    const int w = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 3);
    const int soff = __builtin_amdgcn_readfirstlane((k0 >> 5) * kstep);
    const int nb = __builtin_amdgcn_readfirstlane(nbytes);
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
      const unsigned long long a = (unsigned long long)(uintptr_t)bp[pl];
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((unsigned long long)hi << 32) | lo), 0, nb, 0x00020000);
#pragma unroll
      for (int n = 0; n < NI; ++n) {
        unsigned short* dst = img + pl * BR * BK3 + ((n * 4 + w) * 16) * BK3;       // wave-uniform 1-KiB block
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, voff[n], soff, 0, 0);
      }
    }
  }
  template <int AUX = 0>      // (same call form as Bf3Loader; the builtin itself sits in a non-template member: the host pass of the compiler rejects it inside one)
  __device__ __forceinline__ void issue(int k0, unsigned short* img) const {
    static_assert(AUX == 0, "Bf3BufLoader: default cache policy only");
    issue_plain(k0, img);
  }
};
template <int KIND, int BR, int NPL = 3> struct Bf3LoaderFor { typedef Bf3Loader<KIND, BR, NPL> type; };
template <int BR, int NPL> struct Bf3LoaderFor<OPK_ROWK, BR, NPL> { typedef Bf3BufLoader<BR, NPL> type; };

__device__ __forceinline__ void bf3_lds_read(u32x4& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
}
template <int OFF>
__device__ __forceinline__ void bf3_lds_read_off(u32x4& dst, unsigned addr) {      // OFF: instruction offset, < 65536
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
// planes of the p-th product of a k-step, in the order every kernel of this file adds them: bf16x3 (2,0) (0,2) (1,1) (1,0) (0,1) (0,0);
// f16x2 (1,0) (0,1) (0,0)
constexpr int bf3_prod_plane_a(int npl, int p) { return npl == 3 ? (p == 0 ? 2 : p == 2 || p == 3 ? 1 : 0) : (p == 0 ? 1 : 0); }
constexpr int bf3_prod_plane_b(int npl, int p) { return npl == 3 ? (p == 1 ? 2 : p == 2 || p == 4 ? 1 : 0) : (p == 1 ? 1 : 0); }
// f(std::integral_constant<int, I>) for I = 0 .. N-1, in order (an unrolled loop whose index is a constant expression)
template <int I, int N, class F>
__device__ __forceinline__ void bf3_static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); bf3_static_for<I + 1, N>(f); }
}
typedef float f32x16_ __attribute__((ext_vector_type(16)));
template <int FMT>
__device__ __forceinline__ f32x16_ bf3_mfma(const u32x4& a, const u32x4& b, const f32x16_& c) {
  if constexpr (FMT == 1) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Workgroup tile (64*TM) x (64*TN), 4 waves (2x2), each wave TM x TN MFMA tiles of 32x32.
template <int AK, int TM, int TN, int NSTAGE, int ABL = 0, int FMT = 0>
__global__ void __launch_bounds__(256) gemm_bf3_kernel(const Bf3Params p) {
  constexpr int BM = 64 * TM, BN = 64 * TN, WM = 32 * TM, WN = 32 * TN;
  constexpr int APLANE = BM * BK3, BPLANE = BN * BK3;      // elements per plane image
  constexpr int AOPER = 3 * APLANE, BOPER = 3 * BPLANE;
  constexpr int STAGE = AOPER + BOPER;
  constexpr int NPL = Bf3Fmt<FMT>::NPL;                    // planes in use (the stage keeps three plane slots per operand)
  constexpr int NDMA = NPL * (TM + TN);                    // DMA instructions per wave and K tile
  constexpr int NRD = NPL * (TM + TN);                     // fragment reads per wave and k-step
  __shared__ __align__(1024) unsigned short smem[NSTAGE * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nk = (p.K + BK3 - 1) / BK3;
  int t, kt0 = 0, kt1 = nk, tail_slot = -1;
  if ((int)blockIdx.x < p.tail_first_block) {
    t = xcd_remap(blockIdx.x, p.tail_first_block);
  } else {
    const int q = (int)blockIdx.x - p.tail_first_block;
    const int piece = q % p.tail_split;
    t = p.tail_first_tile + q / p.tail_split;
    const int per = (nk + p.tail_split - 1) / p.tail_split;
    kt0 = piece * per;
    kt1 = min(nk, kt0 + per);
    tail_slot = q;
  }
  const int tm = t / p.ntiles, tn = t - tm * p.ntiles;

  Bf3Loader<AK, BM, NPL> la;
  Bf3Loader<OPK_ROWK, BN, NPL> lbld;
  la.init(p.A, tm * BM, p.M, p.K);
  lbld.init(p.B, tn * BN, p.N, p.K);
  const int nkt = max(kt1 - kt0, 0);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int s0 = 0; s0 < NSTAGE - 1; ++s0)
    if (s0 < nkt) {
      la.issue((kt0 + s0) * BK3, smem + s0 * STAGE);
      lbld.issue((kt0 + s0) * BK3, smem + s0 * STAGE + AOPER);
    }
  const int i31 = lane & 31, h = lane >> 5, key = (i31 >> 2) & 3;
  // byte offsets: row-major 64-B rows, swizzled 16-B position; k-step s uses chunk 2s+h
  const unsigned offA = (unsigned)((wm * WM + i31) * 64), offB = (unsigned)((wn * WN + i31) * 64);
  const unsigned pos[2] = {(unsigned)(((0 + h) ^ key) * 16), (unsigned)(((2 + h) ^ key) * 16)};
  const unsigned sbase0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)smem;

  for (int it = 0; it < nkt; ++it) {
    // tile `it` has landed when at most the (NSTAGE-2) younger tiles' NDMA instructions each are outstanding
    if (NSTAGE == 2 || it + NSTAGE - 2 >= nkt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (NSTAGE - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    const bool more = it + NSTAGE - 1 < nkt;
    unsigned short* nx = smem + ((it + NSTAGE - 1) % NSTAGE) * STAGE;
    const unsigned sb = sbase0 + (unsigned)((it % NSTAGE) * STAGE) * 2u;
    if constexpr (TM == 1 && TN == 1 && FMT == 0) {
      // 64x64 tile: one fused asm block issues all 12 fragment reads (measured ~8 % faster than per-read statements)
      const unsigned a0 = sb + offA, b0 = sb + (unsigned)(AOPER * 2) + offB;
      constexpr unsigned PA = APLANE * 2, PBb = BPLANE * 2;        // plane strides in bytes
      u32x4 ah0, am0, al0, bh0, bm0, bl0, ah1, am1, al1, bh1, bm1, bl1;
      if constexpr (ABL >= 2) {   // ablation: no LDS reads, operands are loop-variant constants
        const unsigned c = 0x3f803f80u + (unsigned)it;
        ah0 = am0 = al0 = bh0 = bm0 = bl0 = ah1 = am1 = al1 = bh1 = bm1 = bl1 = u32x4{c, c, c, c};
        asm volatile("" : "+v"(ah0), "+v"(am0), "+v"(al0), "+v"(bh0), "+v"(bm0), "+v"(bl0));
        asm volatile("" : "+v"(ah1), "+v"(am1), "+v"(al1), "+v"(bh1), "+v"(bm1), "+v"(bl1));
      } else
      asm volatile(
          "ds_read_b128 %0, %12\n\t"
          "ds_read_b128 %3, %14\n\t"
          "ds_read_b128 %1, %12 offset:%c16\n\t"
          "ds_read_b128 %4, %14 offset:%c18\n\t"
          "ds_read_b128 %2, %12 offset:%c17\n\t"
          "ds_read_b128 %5, %14 offset:%c19\n\t"
          "ds_read_b128 %6, %13\n\t"
          "ds_read_b128 %9, %15\n\t"
          "ds_read_b128 %7, %13 offset:%c16\n\t"
          "ds_read_b128 %10, %15 offset:%c18\n\t"
          "ds_read_b128 %8, %13 offset:%c17\n\t"
          "ds_read_b128 %11, %15 offset:%c19\n\t"
          "s_waitcnt lgkmcnt(6)"
          : "=&v"(ah0), "=&v"(am0), "=&v"(al0), "=&v"(bh0), "=&v"(bm0), "=&v"(bl0), "=&v"(ah1), "=&v"(am1),
            "=&v"(al1), "=&v"(bh1), "=&v"(bm1), "=&v"(bl1)
          : "v"(a0 + pos[0]), "v"(a0 + pos[1]), "v"(b0 + pos[0]), "v"(b0 + pos[1]), "i"(PA), "i"(2 * PA), "i"(PBb),
            "i"(2 * PBb)
          : "memory");
      __builtin_amdgcn_sched_barrier(0);
#define DIC_BF3_M1(A_, B_) \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A_), __builtin_bit_cast(bf16x8, B_), acc[0][0], 0, 0, 0);
      DIC_BF3_M1(al0, bh0) DIC_BF3_M1(ah0, bl0) DIC_BF3_M1(am0, bm0)
      if (more && ABL == 0) la.issue((kt0 + it + NSTAGE - 1) * BK3, nx);
      DIC_BF3_M1(am0, bh0) DIC_BF3_M1(ah0, bm0) DIC_BF3_M1(ah0, bh0)
      if (more && ABL == 0) lbld.issue((kt0 + it + NSTAGE - 1) * BK3, nx + AOPER);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah1), "+v"(am1), "+v"(al1), "+v"(bh1), "+v"(bm1), "+v"(bl1)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      DIC_BF3_M1(al1, bh1) DIC_BF3_M1(ah1, bl1) DIC_BF3_M1(am1, bm1)
      DIC_BF3_M1(am1, bh1) DIC_BF3_M1(ah1, bm1) DIC_BF3_M1(ah1, bh1)
#undef DIC_BF3_M1
    } else {
    // fragments of both k-steps, issue order: step 0 (A tiles x planes, B tiles x planes), then step 1
    u32x4 fa[2][TM][3], fb[2][TN][3];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
          bf3_lds_read(fa[ks][i][pl], sb + (unsigned)(pl * APLANE * 2) + offA + (unsigned)(i * 32 * 64) + pos[ks]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
          bf3_lds_read(fb[ks][j][pl], sb + (unsigned)(AOPER * 2 + pl * BPLANE * 2) + offB + (unsigned)(j * 32 * 64) + pos[ks]);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // wait for this k-step's reads (the other step's NRD reads may stay in flight for ks == 0)
      if (ks == 0) {
        if constexpr (NRD <= 15) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NRD) : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) asm volatile("" : "+v"(fa[ks][i][pl]));
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) asm volatile("" : "+v"(fb[ks][j][pl]));
      __builtin_amdgcn_sched_barrier(0);
#define DIC_BF3_MFMA(I_, J_, PA_, PB_) acc[I_][J_] = bf3_mfma<FMT>(fa[ks][I_][PA_], fb[ks][J_][PB_], acc[I_][J_]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          // small terms first: al*bh, ah*bl, am*bm, am*bh, ah*bm, ah*bh   (plane 0 = hi, 1 = mid, 2 = lo)
          if constexpr (FMT == 0) { DIC_BF3_MFMA(i, j, 2, 0) DIC_BF3_MFMA(i, j, 0, 2) DIC_BF3_MFMA(i, j, 1, 1) }
          DIC_BF3_MFMA(i, j, 1, 0) DIC_BF3_MFMA(i, j, 0, 1) DIC_BF3_MFMA(i, j, 0, 0)
          if (ks == 0 && i == 0 && j == 0 && more) {      // next tile's DMA goes out in the MFMA shadow
            la.issue((kt0 + it + NSTAGE - 1) * BK3, nx);
            lbld.issue((kt0 + it + NSTAGE - 1) * BK3, nx + AOPER);
          }
        }
#undef DIC_BF3_MFMA
      __builtin_amdgcn_sched_barrier(0);
    }
      }
  }
  __syncthreads();
  if (tail_slot >= 0) {     // raw partial of this K slice, tile-local [64][64] layout (finished by tail_fixup_kernel)
    if constexpr (TM == 1 && TN == 1) {
      float* dst = p.tail_ws + (long long)tail_slot * 64 * 64;
      const int nl = wn * 32 + (lane & 31), ml = wm * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[(ml + (r & 3) + 8 * (r >> 2)) * 64 + nl] = acc[0][0][r];
    }
    return;
  }
  gemm_epilogue<BM, BN>(p, acc, tm, tn, 0, reinterpret_cast<float*>(smem));
}


#ifdef DIC_EXPERIMENTS
#include "experiments/gemm_bf3_parked.inc"      // parked kernels (pipe / persistent without producer waves / 256x128): experiments build only
#endif


// Few-tiles launches (every one of T output tiles cut into sp K slices, G = T * sp workgroups, no whole tiles): which (tile, slice)
// workgroup b computes, returned as tile * sp + slice.  The plain order - tile = b / sp, slice = b % sp - scatters the workgroups that
// read the same operand panel over all eight XCDs (the hardware deals workgroup b to XCD b % 8), so every XCD's L2 fetches every
// panel: the depth encoder's conv2 weight gradient (36 tiles x 7 slices, K = 30 976) fetched 1.05 GB for 206 MB of operands and ran at
// the memory system's pace (profiles/r04a: 153 us).  Here slice z lives on XCD z: its tiles 0 .. n_z - 1 on that XCD's n_z workgroups,
// all walking the same K range in step, so each panel of the slice crosses the fabric once; the tiles that do not fit (T > n_z) go to
// the workgroups left over (XCDs >= sp, and any beyond T on the first sp).  A bijection for every (T, sp <= 7, G = T * sp).
__device__ __forceinline__ int few_tiles_remap(int b, int T, int sp, int G) {
  if (sp > 7 || G != T * sp) return b;
  const int x = b & 7, i = b >> 3;
  if (x < sp && i < T) return i * sp + x;
  int rank = 0;                                   // this workgroup's rank among the left-over workgroups, XCD-major
  for (int xx = 0; xx < 8; ++xx) {
    const int n = (G - xx + 7) >> 3, first = xx < sp ? min(n, T) : 0;
    if (xx == x) { rank += i - first; break; }
    rank += n - first;
  }
  for (int z = 0; z < sp; ++z) {                  // the rank-th left-over (slice, tile): slice-major
    const int f = min((G - z + 7) >> 3, T), cnt = T - f;
    if (rank < cnt) return (f + rank) * sp + z;
    rank -= cnt;
  }
  return b;
}

// Warp-specialised form of gemm_bf3_persist_kernel: waves 0..3 only compute (fragment reads, MFMAs, stores), waves 4..7 only
// move operands (LDS-DMA issue and the counted vmcnt that says a slot has landed); both meet at the one barrier per K tile.
// Why: with DMA issue inside the computing waves a K tile costs 3500-4000 cycles against 2500-2600 without any DMA
// (scripts/bench_bf3_pipe_ablate.py) - a global_load_lds that cannot issue (address arithmetic, M0 set-up, a full
// vector-memory queue) blocks the MFMAs queued behind it in the same wave.  A producer wave that blocks costs nothing.
// Two waves per SIMD, so the kernel has to fit 256 registers; the stores of a seam and the DMA no longer share a vmcnt.
// NPW (on-the-fly operand only): producer waves, 4 or 8.  With four, a producer thread transforms 16 values per K tile - ~150 dependent
// vector instructions between "the slot's data has arrived" and "its LDS image is written", longer than the computing wave's 24 MFMAs
// of the same K tile: the producer, not the matrix pipe, set the pace (1.34 us per K tile against 0.83 us for the plane kernel).  Eight
// producer waves (two per SIMD beside the computing wave: 768 threads, <= 168 registers) halve that chain and overlap two of them per SIMD.
// ILV (round 4, default): the computing waves issue one fragment read in the gap behind each matrix instruction instead of a block of reads
// in front of each k-step's matrix instructions (see conv3x3_bf3_halo_kernel); same products in the same order, bit-identical to ILV = 0.
template <int AK, int ABL = 0, int NST = 3, int FMT = 0, int NPW = 4, int DA_ = 4, int ILV = 1>      // DA_: input slots in flight per producer wave (on-the-fly operand); FMT: operand format (top of the file); ABL (measurement only): 1 = the producer waves issue nothing inside the loop, 3 = A always re-fetches K tile 0 of its output tile (cache-hot A), 4 = A and B both; NST: ring stages
__global__ void __launch_bounds__(256 + 64 * NPW) gemm_bf3_persist_ws_kernel(const Bf3Params p) {
  constexpr int BM = 128, BN = 128;
  static_assert(NPW == 4 || (NPW == 8 && AK == OPK_ROWK_BN), "eight producer waves: on-the-fly operand only");
  constexpr int NPL = Bf3Fmt<FMT>::NPL;       // planes per operand
  constexpr int APLANE = BM * BK3, BPLANE = BN * BK3, AOPER = NPL * APLANE, BOPER = NPL * BPLANE, STAGE = AOPER + BOPER;
  __shared__ __align__(1024) unsigned short smem[NST * STAGE];
  constexpr bool kBn = AK == OPK_ROWK_BN;
  __shared__ float bn_tab[kBn ? 2 * kBnTabMax : 1];      // scale | shift of the on-the-fly operand (144 + 16 KB = all of the LDS)
  __shared__ float res_tab[(kBn && FMT == 1) ? 2 * kBnTabMax : 1];      // f16x2 (96 + 16 + 16 KB): scale | shift of the residual's own BatchNorm

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nkt = (p.K + BK3 - 1) / BK3;
  const int T = p.mtiles * p.ntiles, G = gridDim.x;
  // Work list of this workgroup: its whole tiles xcd_remap(blockIdx + j*G, F), j < ntl, then - remainder-round K split,
  // tail_split > 1 - at most ONE piece: K tiles [pk0, pk0 + npk) of remainder tile F + blockIdx / tail_split.  The T - F
  // remainder tiles (less than one round of the grid) are cut into tail_split K slices so that every CU gets a share of the
  // last round; a piece leaves its raw accumulators in tail_ws (one [64][64] slab per consumer wave = per 64x64 quadrant)
  // and tail_fixup_kernel (gemm.hip, 128x128 mode) sums the slices in K order, stores, and writes the BatchNorm partials.
  const bool split = p.tail_split > 1;
  const int F = split ? p.tail_first_tile : T;
  const int ntl = (F - (int)blockIdx.x + G - 1) / G;
  const int per = split ? (nkt + p.tail_split - 1) / p.tail_split : 0;
  const int pb = (split && F == 0 && p.few_remap) ? few_tiles_remap((int)blockIdx.x, T, p.tail_split, G) : (int)blockIdx.x;      // piece index (few-tiles launches: XCD-aware)
  const bool has_piece = split && pb < (T - F) * p.tail_split;
  const int piece_tile = has_piece ? F + pb / p.tail_split : 0;
  const int pk0 = has_piece ? (pb % p.tail_split) * per : 0;
  const int npk = has_piece ? min(nkt, pk0 + per) - pk0 : 0;
  const int nwork = ntl + (has_piece ? 1 : 0);
  const int total = ntl * nkt + npk;
  if (total == 0) return;
  if constexpr (kBn) {     // scale / shift of every input channel -> LDS (all eight waves; one barrier, once per launch)
    for (int k = tid; k < p.K; k += 256 + 64 * NPW) { bn_tab[k] = p.a_scale[k]; bn_tab[kBnTabMax + k] = p.a_shift[k]; }
    if constexpr (FMT == 1) {
      if (p.a_res_scale)
        for (int k = tid; k < p.K; k += 256 + 64 * NPW) { res_tab[k] = p.a_res_scale[k]; res_tab[kBnTabMax + k] = p.a_res_shift[k]; }
    }
    __syncthreads();
  }

  if constexpr (kBn) {
  if (wave >= 4) {
    // ---------------- producer waves, on-the-fly A operand.  B tiles: LDS-DMA three slots ahead, as in the plain kernel.  A tiles:
    // every thread loads 16 raw fp32 values (and residuals) of the 128 x 32 tile into registers two slots ahead,
    // and one slot ahead of the consumers turns them into act(raw * scale + shift (+ residual)), splits them into the three
    // bf16 planes (v_cvt_pk_bf16_f32: the same round-to-nearest-even as split3_bf16) and writes the swizzled LDS image the
    // DMA would have produced.  Slot s lives in register set s % DA.
    //   iteration g:  wait L(g+1), D(g+1)  ->  transform slot g+1 into stage (g+1) % 3 [+ store the fp32 values]  ->  barrier g
    //                 ->  D(g+3) into stage g % 3,  L(g+1+DA) into the register set just freed
    // (a load has DA - 1 K-tile periods to arrive: with two sets the kernel ran at the memory latency, 2.5 us per K tile)
    // Thread mapping of the 128 x 32 fp32 tile: a wave instruction reads 8 rows x 128 B (8 lanes x 16 B per row: whole cache
    // lines, every byte used once); thread (r8, kq) of producer wave pw holds channels 4*kq..4*kq+3 of rows pw*32 + 8*i + r8.
    // (NPW = 8: 16 rows per producer wave, two rows per thread; only the first four producer waves issue the weight DMA)
    constexpr int RPW = BM / NPW, NR = RPW / 8;      // rows per producer wave, rows per thread
    const int pt = tid - 256, prow0 = (pt >> 6) * RPW + ((pt & 63) >> 3), kq = pt & 7;
    const bool b_wave = (pt >> 6) < 4;               // wave-uniform
    __builtin_amdgcn_s_setprio(3);       // the transform's vector instructions ahead of the computing wave of the same SIMD (measured: no change
                                         // either way - the two instruction streams add up on the SIMD whatever their order)
    const bool has_res = p.a_res != nullptr;
    const bool has_res_bn = has_res && p.a_res_scale != nullptr;
    typename Bf3LoaderFor<OPK_ROWK, BN, NPL>::type lbld;
    // ---- slot iterators: (work item, K tile) of the next B slot to issue / next A slot to load / next A slot to transform
    struct It { int j, kt, k0, nk; };
    auto it_init = [&](It& it) { it.j = 0; it.kt = 0; it.k0 = ntl > 0 ? 0 : pk0; it.nk = ntl > 0 ? nkt : npk; };
    auto it_tile = [&](const It& it) { return it.j < ntl ? xcd_remap(blockIdx.x + it.j * G, F) : piece_tile; };
    auto it_next = [&](It& it) {          // returns true when it moved on to the next work item
      if (++it.kt < it.nk) return false;
      it.kt = 0; ++it.j;
      const bool whole = it.j < ntl;
      it.k0 = whole ? 0 : pk0; it.nk = whole ? nkt : npk;
      return true;
    };
    It itb, itl;
    it_init(itb); it_init(itl);
    if (b_wave) { const int t = it_tile(itb); lbld.init(p.B, (t % p.ntiles) * BN, p.N, p.K); }
    auto issue_b = [&](unsigned short* stage) {
      if (!b_wave) return;
      lbld.issue((itb.k0 + itb.kt) * BK3, stage + AOPER);
      if (it_next(itb) && itb.j < nwork) { const int t = it_tile(itb); lbld.init(p.B, (t % p.ntiles) * BN, p.N, p.K); }
    };
    // register set of one slot (DA instances; indices are compile-time constants after unrolling: no runtime-indexed register arrays)
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    struct ASet {
      f32x4_ ra[NR], rr[NR];              // raw values, residuals (row prow0 + 8 * i)
      unsigned off0;                      // byte offset of the thread's first row in a_raw / a_res / a_out
      unsigned tab;                       // byte offset of the thread's first channel in the scale / shift table
      unsigned flg;                       // bits 0..3: row i inside the matrix, bit 4: this tile stores a_out, bit 5: ragged tile (wave-uniform)
    };
    constexpr int DA = DA_;               // A slots in flight (register sets); slot s lives in set s % DA
    ASet sets[DA];
    if (!has_res) {                       // the residual registers stay zero for the whole launch (the transform always adds them)
#pragma unroll
      for (int d = 0; d < DA; ++d)
#pragma unroll
        for (int i = 0; i < NR; ++i) sets[d].rr[i] = f32x4_{0.f, 0.f, 0.f, 0.f};
    }
    // loader state of the tile `itl` is in: 32-bit byte offsets (the host checks M * ld * 4 < 2^32) of the thread's four rows
    // (clamped to the last row of the matrix) at channel 4 * kq, and the flags every slot of the tile carries
    const unsigned ldb = (unsigned)p.a_ld * 4u;
    unsigned lrow[NR], lrow0 = 0, lflg = 0;
    auto load_tile = [&]() {
      const int t = it_tile(itl), tm = t / p.ntiles, tn = t - tm * p.ntiles;
      const int gr = tm * BM + prow0;
      lrow0 = (unsigned)gr * ldb + (unsigned)kq * 16u;
      // the fp32 copy of the input (a_out) is written once per row block: by the N tile that "owns" this producer wave's 32 rows - the
      // four waves are dealt over the first min(ntiles, 4) N tiles of the row block, so that sibling tiles carry the same store load and
      // stay in step (they read the same input rows: in step, the second read hits L2; until round 4 the tn == 0 tile stored everything)
      const int owner = ((prow0 >> 5) * min(p.ntiles, 4)) >> 2;      // (prow0 >> 5: the 32-row quarter of the row block this thread works in)
      lflg = ((tn == owner && p.a_out != nullptr) ? 16u : 0u) | ((tm * BM + BM > p.M) ? 32u : 0u);
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        lflg |= (gr + 8 * i < p.M) ? (1u << i) : 0u;
        lrow[i] = (unsigned)min(gr + 8 * i, p.M - 1) * ldb + (unsigned)kq * 16u;
      }
    };
    load_tile();
    auto load_a = [&](ASet& S) {          // L(s): the slot at `itl`
      const unsigned kb = (unsigned)(itl.k0 + itl.kt) * (BK3 * 4u);
      S.off0 = lrow0 + kb; S.tab = kb + (unsigned)kq * 16u; S.flg = lflg;
      // (inline asm: a load the compiler can see gets a compiler-placed vmcnt(0) at its first use - it cannot count across this
      //  control flow - which drains every slot in flight; the counted waits below are followed by a statement naming the registers)
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const unsigned vo = lrow[i] + kb;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(S.ra[i]) : "v"(vo), "s"(p.a_raw) : "memory");
      }
      if (has_res) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const unsigned vo = lrow[i] + kb;
          asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(S.rr[i]) : "v"(vo), "s"(p.a_res) : "memory");
        }
      }
      if (it_next(itl) && itl.j < nwork) load_tile();
    };
    // LDS image: row r of a plane is 64 B (32 bf16); its 16-B chunk c sits at position c ^ ((r >> 2) & 3); the thread's 4 values
    // are the 8-B half (kq & 1) of chunk kq >> 1
    unsigned doff[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const unsigned r = (unsigned)(prow0 + 8 * i);
      doff[i] = r * 64u + ((((unsigned)kq >> 1) ^ ((r >> 2) & 3u)) << 4) + ((unsigned)kq & 1u) * 8u;
    }
    const float relu_floor = p.a_relu ? 0.f : -__builtin_inff();
    auto transform = [&](ASet& S, unsigned short* stage) {      // T(s): registers -> three plane images of the stage
      // (LDS accesses of the producer waves are inline asm with their own lgkmcnt waits: hipcc would otherwise drain every
      //  LDS-DMA in flight - vmcnt(0) - before an LDS access it can see)
      u32x4 scq, shq;
      const unsigned tab = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)bn_tab + S.tab;
      bf3_lds_read(scq, tab); bf3_lds_read(shq, tab + (unsigned)kBnTabMax * 4u);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(scq), "+v"(shq)::"memory");
      const float4 s4 = __builtin_bit_cast(float4, scq), t4 = __builtin_bit_cast(float4, shq);
      const unsigned sbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)stage;
      typedef __bf16 bfx2 __attribute__((ext_vector_type(2)));
      typedef float f32x2_ __attribute__((ext_vector_type(2)));
      typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
      float v[NR][4];
      const f32x2_ s01 = {s4.x, s4.y}, s23 = {s4.z, s4.w}, t01 = {t4.x, t4.y}, t23 = {t4.z, t4.w};
      if constexpr (FMT == 1) {
        if (has_res_bn) {      // the residual's own BatchNorm first (wave-uniform branch): q <- fma(q, scale, shift), rounded to fp32
          u32x4 rsq, rtq;
          const unsigned rtab = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)res_tab + S.tab;
          bf3_lds_read(rsq, rtab); bf3_lds_read(rtq, rtab + (unsigned)kBnTabMax * 4u);
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rsq), "+v"(rtq)::"memory");
          const float4 a4 = __builtin_bit_cast(float4, rsq), b4 = __builtin_bit_cast(float4, rtq);
#pragma unroll
          for (int i = 0; i < NR; ++i) {
            S.rr[i].x = fmaf(S.rr[i].x, a4.x, b4.x); S.rr[i].y = fmaf(S.rr[i].y, a4.y, b4.y);
            S.rr[i].z = fmaf(S.rr[i].z, a4.z, b4.z); S.rr[i].w = fmaf(S.rr[i].w, a4.w, b4.w);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const f32x4_ x = S.ra[i], q = S.rr[i];
        const f32x2_ a01 = f32x2_{x.x, x.y} * s01 + t01 + f32x2_{q.x, q.y}, a23 = f32x2_{x.z, x.w} * s23 + t23 + f32x2_{q.z, q.w};    // packed fp32 fma / add
        v[i][0] = fmaxf(a01.x, relu_floor); v[i][1] = fmaxf(a01.y, relu_floor);
        v[i][2] = fmaxf(a23.x, relu_floor); v[i][3] = fmaxf(a23.y, relu_floor);
      }
      if constexpr (FMT == 1) {
        // overflow guard (common.h): the values are >= 0 here (a_relu) or at least no NaN survives fmaxf, so the largest of the
        // sixteen decides - 8 vector instructions and a branch per slot instead of a compare per value
        float mx = fabsf(v[0][0]);
#pragma unroll
        for (int i = 0; i < NR; ++i) mx = fmaxf(fmaxf(fmaxf(mx, fabsf(v[i][0])), fmaxf(fabsf(v[i][1]), fabsf(v[i][2]))), fabsf(v[i][3]));
        if (mx > kF16Max / kF16ActScale) f16x2_raise(p.status, 4u);
      }
      if (__builtin_amdgcn_readfirstlane(S.flg) & 32u) {        // last M tile of a ragged matrix: rows past the end are zero
        asm volatile("" ::: "memory");                           // (keeps this a branch: 16 selects per slot otherwise)
#pragma unroll
        for (int i = 0; i < NR; ++i)
          if (!((S.flg >> i) & 1u)) { v[i][0] = 0.f; v[i][1] = 0.f; v[i][2] = 0.f; v[i][3] = 0.f; }
      }
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        if ((S.flg & 16u) && ((S.flg >> i) & 1u))
          *reinterpret_cast<float4*>(reinterpret_cast<char*>(p.a_out) + (size_t)(S.off0 + (unsigned)(8 * i) * ldb)) =
              make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
        const unsigned dst = sbase + doff[i];
        if constexpr (FMT == 1) {         // two fp16 planes of kF16ActScale * v
          typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
          u32x2_ q1, q2;
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const f32x2_ xx = f32x2_{v[i][2 * u], v[i][2 * u + 1]} * kF16ActScale;
            const f16x2_ h1 = __builtin_convertvector(xx, f16x2_);
            const f32x2_ r1 = xx - __builtin_convertvector(h1, f32x2_);
            q1[u] = __builtin_bit_cast(unsigned, h1); q2[u] = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, f16x2_));
          }
          asm volatile("ds_write_b64 %0, %1" ::"v"(dst), "v"(q1) : "memory");
          asm volatile("ds_write_b64 %0, %1 offset:%c2" ::"v"(dst), "v"(q2), "i"(APLANE * 2) : "memory");
        } else {
        u32x2_ qh, qm, ql;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const f32x2_ xx = {v[i][2 * u], v[i][2 * u + 1]};
          const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(xx, bfx2));
          const f32x2_ r1 = {xx.x - __uint_as_float(hb << 16), xx.y - __uint_as_float(hb & 0xffff0000u)};
          const unsigned mb = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bfx2));
          const f32x2_ r2 = {r1.x - __uint_as_float(mb << 16), r1.y - __uint_as_float(mb & 0xffff0000u)};
          qh[u] = hb; qm[u] = mb; ql[u] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bfx2));
        }
        asm volatile("ds_write_b64 %0, %1" ::"v"(dst), "v"(qh) : "memory");
        asm volatile("ds_write_b64 %0, %1 offset:%c2" ::"v"(dst), "v"(qm), "i"(APLANE * 2) : "memory");
        asm volatile("ds_write_b64 %0, %1 offset:%c2" ::"v"(dst), "v"(ql), "i"(2 * APLANE * 2) : "memory");
        }
      }
    };
    // ---- prologue: B slots 0..2 and A slots 0..DA-1 in flight; slot 0 transformed; then A slot DA
#pragma unroll
    for (int s0 = 0; s0 < NST; ++s0)
      if (s0 < total) issue_b(smem + s0 * STAGE);
#pragma unroll
    for (int s0 = 0; s0 < DA; ++s0)
      if (s0 < total) load_a(sets[s0]);
    const bool steady_ok = total >= 2 * DA + 2;                    // shorter streams: plain vmcnt(0) waits
    // The wait names no registers (s_waitcnt takes an immediate, picked by scalar branches); ONE statement after it names the
    // registers of the slot just released: every use of them is ordered behind it.  (A wait with the registers as operands in each
    // branch arm made the compiler copy the whole set - before the wait, i.e. before the data had arrived.)
#define DIC_PIN_SLOT(S_)                                                                                                          \
  do { _Pragma("unroll") for (int i_ = 0; i_ < NR; ++i_) asm volatile("" : "+v"((S_).ra[i_]), "+v"((S_).rr[i_]) :: "memory"); } while (0)
    // ---- prologue (drains once per launch): slot 0 transformed, its register set reloaded
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DIC_PIN_SLOT(sets[0]);
    transform(sets[0], smem);
    if (DA < total) load_a(sets[0]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                  // slot 0 is in LDS
    int st = 0;
    for (int g0 = 0; g0 < total; g0 += DA) {
#pragma unroll
      for (int u = 0; u < DA; ++u) {
        const int g = g0 + u;
        if (g >= total) break;
        const int stn = st == NST - 1 ? 0 : st + 1;
        // iteration g:  D(g+1) and L(g+1) done  ->  T(g+1)  ->  barrier g  ->  D(g+3) into the stage just freed, L(g+1+DA) into the
        // register set just freed.  D(g+1) was issued in iteration g-2, L(g+1) long before it; younger than D(g+1) in issue order:
        // L(g-1+DA), D(g+2), L(g+DA) [and stores of tiles that keep the fp32 copy - not counted: the wait then covers more, never
        // less].  At the start and the end of the stream, where that pattern is incomplete, wait for everything.
        // (Until this build the count was (DA-1) * (2*NPL + nl) - enough for L(g+1) but not for the weight tile D(g+1), which is
        //  much younger; no test ever caught a late tile, the weights come from L2 within two K tiles, but nothing guaranteed it.)
        if (g + 1 < total) {
          if (steady_ok && g >= 2 && g + DA < total) {
            // (waves without the weight DMA - NPW = 8 - only need L(g+1): the DA - 1 younger loads may stay in flight)
            // (stores of the fp32 copy, tiles that keep it, are younger than the awaited slot and not counted: the wait then covers more, never
            //  less.  Counting them exactly - round 4, a history bit per transform - changed nothing: 54.0 us inside a ResNet forward either way)
            if (b_wave) { if (has_res) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * 2 * NR + 2 * NPL) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NR + 2 * NPL) : "memory"); }
            else { if (has_res) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DA - 1) * 2 * NR) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DA - 1) * NR) : "memory"); }
          } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          DIC_PIN_SLOT(sets[(u + 1) % DA]);
          transform(sets[(u + 1) % DA], smem + stn * STAGE);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                              // slot g+1 is in LDS; the consumers are done with stage st
        if (g + NST < total) issue_b(smem + st * STAGE);
        if (g + 1 + DA < total) load_a(sets[(u + 1) % DA]);
        st = stn;
      }
    }
#undef DIC_PIN_SLOT
    return;
  }
  } else
  if (wave >= 4) {
    // ---------------- producer waves: slot g of the stream goes to ring stage g % NST
    typename Bf3LoaderFor<AK, BM, NPL>::type la;
    typename Bf3LoaderFor<OPK_ROWK, BN, NPL>::type lbld;
    int pj = 0, pkt = 0, k0cur = ntl > 0 ? 0 : pk0, nkcur = ntl > 0 ? nkt : npk;
    {
      const int t = ntl > 0 ? xcd_remap(blockIdx.x, F) : piece_tile;
      la.init(p.A, (t / p.ntiles) * BM, p.M, p.K);
      lbld.init(p.B, (t % p.ntiles) * BN, p.N, p.K);
    }
    auto prefetch = [&](unsigned short* stage) {
      la.template issue<DIC_WS_A_AUX>((ABL >= 3 ? 0 : k0cur + pkt) * BK3, stage);
      lbld.issue((ABL >= 4 ? 0 : k0cur + pkt) * BK3, stage + AOPER);
      if (++pkt == nkcur) {
        pkt = 0; ++pj;
        if (pj < nwork) {
          const bool whole = pj < ntl;
          const int t = whole ? xcd_remap(blockIdx.x + pj * G, F) : piece_tile;
          k0cur = whole ? 0 : pk0; nkcur = whole ? nkt : npk;
          la.init(p.A, (t / p.ntiles) * BM, p.M, p.K);
          lbld.init(p.B, (t % p.ntiles) * BN, p.N, p.K);
        }
      }
    };
#pragma unroll
    for (int s0 = 0; s0 < NST; ++s0)
      if (s0 < total) prefetch(smem + s0 * STAGE);
    if (total >= NST) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NPL * (NST - 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                  // slot 0 is in LDS
    int st = 0;
    for (int g = 0; g < total; ++g) {
      // before the consumers read slot g+1 (after this barrier) it must have landed; slot g+2 may stay in flight
      if (NST >= 3 && g + 2 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NPL) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!(ABL == 6 && (g & 1))) __builtin_amdgcn_s_barrier();    // ... and the consumers are done with stage st  (ABL 6, timing only: every other barrier skipped)
      if (ABL != 1 && g + NST < total) prefetch(smem + st * STAGE);
      st = st == NST - 1 ? 0 : st + 1;
    }
    return;
  }

  // ---------------- consumer waves
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int i31 = lane & 31, h = lane >> 5, key = (i31 >> 2) & 3;
  const unsigned offA = (unsigned)((wm * 64 + i31) * 64), offB = (unsigned)((wn * 64 + i31) * 64);
  const unsigned pos[2] = {(unsigned)(((0 + h) ^ key) * 16), (unsigned)(((2 + h) ^ key) * 16)};
  const unsigned sbase0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)smem;
  u32x4 fa[2][2][3], fb[2][2][3];
#define DIC_PIPE_READ_A(KS_, SB_, I_)                                                                                \
  _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl)                                                                 \
      bf3_lds_read(fa[KS_][I_][pl], (SB_) + (unsigned)(pl * APLANE * 2) + offA + (unsigned)((I_) * 32 * 64) + pos[KS_]);
#define DIC_PIPE_READ_B(KS_, SB_, J_)                                                                                \
  _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl)                                                                 \
      bf3_lds_read(fb[KS_][J_][pl], (SB_) + (unsigned)(AOPER * 2 + pl * BPLANE * 2) + offB + (unsigned)((J_) * 32 * 64) + pos[KS_]);
#define DIC_PIPE_PIN(KS_)                                                                                            \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl) {                 \
    asm volatile("" : "+v"(fa[KS_][i][pl])); asm volatile("" : "+v"(fb[KS_][i][pl])); }
#define DIC_PIPE_MFMA(KS_, PA_, PB_)                                                                                 \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                        \
      acc[i][j] = bf3_mfma<FMT>(fa[KS_][i][PA_], fb[KS_][j][PB_], acc[i][j]);
  __builtin_amdgcn_s_barrier();                                    // slot 0 is in LDS
  DIC_PIPE_READ_A(0, sbase0, 0) DIC_PIPE_READ_A(0, sbase0, 1) DIC_PIPE_READ_B(0, sbase0, 0) DIC_PIPE_READ_B(0, sbase0, 1)
  int g = 0, st = 0;
  for (int j = 0; j < nwork; ++j) {
    const int nkj = j < ntl ? nkt : npk;
    for (int kt = 0; kt < nkj; ++kt, ++g) {
      const int stn = st == NST - 1 ? 0 : st + 1;
      const unsigned sb = sbase0 + (unsigned)(st * STAGE) * 2u, sbn = sbase0 + (unsigned)(stn * STAGE) * 2u;
      if constexpr (ILV != 0 && ABL != 6) {
        constexpr int NP = NPL == 3 ? 6 : 3;
        // ---- k-step 0 (its fragments were requested during the previous slot's second half); k-step 1's fragments in the gaps
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        DIC_PIPE_PIN(0)
        __builtin_amdgcn_sched_barrier(0);
        const unsigned a1 = sb + offA + pos[1], b1 = sb + (unsigned)(AOPER * 2) + offB + pos[1];
        bf3_static_for<0, 4 * NP>([&](auto mc) {
          constexpr int m = decltype(mc)::value, ai = (m & 3) >> 1, aj = m & 1;
          acc[ai][aj] = bf3_mfma<FMT>(fa[0][ai][bf3_prod_plane_a(NPL, m >> 2)], fb[0][aj][bf3_prod_plane_b(NPL, m >> 2)], acc[ai][aj]);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (m < 2 * NPL) {
            bf3_lds_read_off<(m % NPL) * APLANE * 2 + (m / NPL) * 32 * 64>(fa[1][m / NPL][m % NPL], a1);
            __builtin_amdgcn_sched_barrier(0);
          } else if constexpr (m < 4 * NPL) {
            bf3_lds_read_off<(m % NPL) * BPLANE * 2 + ((m - 2 * NPL) / NPL) * 32 * 64>(fb[1][(m - 2 * NPL) / NPL][m % NPL], b1);
            __builtin_amdgcn_sched_barrier(0);
          }
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // every fragment of this stage is in registers
        DIC_PIPE_PIN(1)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- k-step 1; k-step 0 of the next slot in the gaps (past the last slot: a harmless re-read of a ring stage)
        const unsigned a0 = sbn + offA + pos[0], b0 = sbn + (unsigned)(AOPER * 2) + offB + pos[0];
        bf3_static_for<0, 4 * NP>([&](auto mc) {
          constexpr int m = decltype(mc)::value, ai = (m & 3) >> 1, aj = m & 1;
          acc[ai][aj] = bf3_mfma<FMT>(fa[1][ai][bf3_prod_plane_a(NPL, m >> 2)], fb[1][aj][bf3_prod_plane_b(NPL, m >> 2)], acc[ai][aj]);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (m < 2 * NPL) {
            bf3_lds_read_off<(m % NPL) * APLANE * 2 + (m / NPL) * 32 * 64>(fa[0][m / NPL][m % NPL], a0);
            __builtin_amdgcn_sched_barrier(0);
          } else if constexpr (m < 4 * NPL) {
            bf3_lds_read_off<(m % NPL) * BPLANE * 2 + ((m - 2 * NPL) / NPL) * 32 * 64>(fb[0][(m - 2 * NPL) / NPL][m % NPL], b0);
            __builtin_amdgcn_sched_barrier(0);
          }
        });
        st = stn;
        continue;
      }
      DIC_PIPE_READ_A(1, sb, 0) DIC_PIPE_READ_A(1, sb, 1) DIC_PIPE_READ_B(1, sb, 0) DIC_PIPE_READ_B(1, sb, 1)
      if constexpr (NPL == 3) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
      DIC_PIPE_PIN(0)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NPL == 3) { DIC_PIPE_MFMA(0, 2, 0) DIC_PIPE_MFMA(0, 0, 2) DIC_PIPE_MFMA(0, 1, 1) }
      DIC_PIPE_MFMA(0, 1, 0) DIC_PIPE_MFMA(0, 0, 1) DIC_PIPE_MFMA(0, 0, 0)
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // every fragment of this stage is in registers
      DIC_PIPE_PIN(1)
      if (!(ABL == 6 && (g & 1))) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NPL == 3) { DIC_PIPE_MFMA(1, 2, 0) } else { DIC_PIPE_MFMA(1, 1, 0) }
      if (g + 1 < total) { DIC_PIPE_READ_A(0, sbn, 0) DIC_PIPE_READ_A(0, sbn, 1) DIC_PIPE_READ_B(0, sbn, 0) DIC_PIPE_READ_B(0, sbn, 1) }
      if constexpr (NPL == 3) { DIC_PIPE_MFMA(1, 0, 2) DIC_PIPE_MFMA(1, 1, 1) DIC_PIPE_MFMA(1, 1, 0) }
      DIC_PIPE_MFMA(1, 0, 1) DIC_PIPE_MFMA(1, 0, 0)
      __builtin_amdgcn_sched_barrier(0);
      st = stn;
    }
    // ---- seam: store the tile.  The piece of a remainder tile (j == ntl) goes through the same stores with the output
    // matrix replaced by this wave's [64][64] slab of raw partial sums in tail_ws (quadrant-major, then slice): no bias /
    // activation / statistics - tail_fixup_kernel sums the slices and does the rest
    const bool piece = j >= ntl;
    const int t = piece ? piece_tile : xcd_remap(blockIdx.x + j * G, F);
    const int tm = t / p.ntiles, tn = t - tm * p.ntiles;
    const bool full = piece || ((tm + 1) * BM <= p.M && (tn + 1) * BN <= p.N);
    const bool plain = piece || (!p.ep.bias && p.ep.act == ACT_NONE && !p.ep.accumulate);
    const int n0 = piece ? (lane & 31) : tn * BN + wn * 64 + (lane & 31), m0 = piece ? 4 * h : tm * BM + wm * 64 + 4 * h;
    float* const Cb = piece ? p.tail_ws + ((long long)((pb / p.tail_split) * 4 + wave) * p.tail_split + pb % p.tail_split) * (64 * 64)
                            : p.ep.C;
    const long long ldc = piece ? 64 : p.ep.ldc;
    float cs[2] = {0.f, 0.f}, cs2[2] = {0.f, 0.f};
    if constexpr (FMT == 1) {             // undo the operands' power-of-two scales (a piece stays raw: the fix-up scales the sum)
      if (!piece) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) acc[i][jj] *= ep_alpha(p.ep);
      }
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float* col = Cb + (long long)(m0 + i * 32) * ldc + n0 + jj * 32;
        if (!plain) {      // bias / activation / accumulate (finalize_store's order): the stored value replaces the accumulator
          const int n = n0 + jj * 32;
          const float bcol = (p.ep.bias && n < p.N) ? p.ep.bias[n] : 0.f;
          float old[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {        // C += result: all 16 old values in flight before the first store
            const bool in = m0 + i * 32 + (r & 3) + 8 * (r >> 2) < p.M && n < p.N;
            old[r] = (p.ep.accumulate && in) ? col[(long long)((r & 3) + 8 * (r >> 2)) * ldc] : 0.f;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[i][jj][r] + bcol;
            if (p.ep.act == ACT_RELU) v = fmaxf(v, 0.f);
            else if (p.ep.act == ACT_SIGMOID) v = sigmoidf_(v);
            else if (p.ep.act == ACT_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
            const bool in = m0 + i * 32 + (r & 3) + 8 * (r >> 2) < p.M && n < p.N;
            v = in ? v + old[r] : 0.f;          // outside the matrix: nothing stored, nothing in the statistics
            if (in) col[(long long)((r & 3) + 8 * (r >> 2)) * ldc] = v;
            acc[i][jj][r] = v;
          }
        } else if (full) {
#pragma unroll
          for (int r = 0; r < 16; ++r) col[(long long)((r & 3) + 8 * (r >> 2)) * ldc] = acc[i][jj][r];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (m0 + i * 32 + (r & 3) + 8 * (r >> 2) < p.M && n0 + jj * 32 < p.N)
              col[(long long)((r & 3) + 8 * (r >> 2)) * ldc] = acc[i][jj][r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {       // rows beyond M hold exact zeros (zero-filled operand rows)
          cs[jj] += acc[i][jj][r]; cs2[jj] += acc[i][jj][r] * acc[i][jj][r];
          acc[i][jj][r] = 0.f;
        }
      }
    if (p.ep.stats && !piece) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const float a = cs[jj] + __shfl_xor(cs[jj], 32, 64), b = cs2[jj] + __shfl_xor(cs2[jj], 32, 64);
        const int n = n0 + jj * 32;
        if (lane < 32 && n < p.N) {
          p.ep.stats[((long long)(tm * 2 + wm) * 2 + 0) * p.N + n] = a;
          p.ep.stats[((long long)(tm * 2 + wm) * 2 + 1) * p.N + n] = b;
        }
      }
    }
  }
#undef DIC_PIPE_READ_A
#undef DIC_PIPE_READ_B
#undef DIC_PIPE_PIN
#undef DIC_PIPE_MFMA
}

// 256x128 form of the warp-specialised persistent kernel, for grids deep enough to fill the CUs with half as many tiles: eight
// computing waves (4 x 2, 64x64 each) + four producer waves = three waves per SIMD, 168 registers each.  History: with the
// pointer-form DMA and bf16x3 operands it lost 4-7 % against the 128x128 form (203 -> 216 us on 50176x256x1024) and was parked;
// with the buffer-load DMA (no vector instructions in the producer's K loop) and the f16x2 format (half the matrix work per K
// tile, so the second computing wave of a SIMD has latency to hide) it wins 15-19 % on the 1x1 expansions (layer-3 conv3
// 34.4 -> 28.0 us, layer-2 conv3 41.1 -> 34.1) and is the f16x2 kernel for plain-epilogue row-major launches of >= 192 such tiles.
// Operand bytes per MFMA drop by a quarter (72 KB per K tile for twice the MFMAs), which is what the 128x128 form still waits
// for (scripts/bench_bf3_ws_ablate.py: 3100-3900 cycles per K tile against 2400-2600 without any DMA).  Two ring stages of
// 72 KB; B fragments are single-buffered (the second computing wave of the SIMD covers their latency), A fragments stay
// double-buffered by k-step.  BatchNorm partials per 64-row wave tile: [4*mtiles][2][N].  Bit-identical to gemm_bf3_kernel.
// BNA (round 4, f16x2 only; PARKED - instantiated in the experiments build only, switch 107): built, bit-identical to the plane route on
// every conv3 shape of the network (scripts/experiments/test_parked_kernels_gpu.py), and neutral: 35.3 us per layer-3 launch against
// 28.6 + 6.3 us for the plane kernel and the bn_apply_planes pass it replaces, pipelined step 9.03 ms with conv3 folded onto it and
// 9.03 ms without (the 12.8-MB tensors between conv2 and conv3 never leave the Infinity Cache: removing their passes removes no HBM
// traffic - unlike the 51-MB block boundary, where bytes came off the step one for one).
// The A operand is formed on the fly, A(m,k) = act(a_raw[m][k] * a_scale[k] + a_shift[k]) - conv3 reading the
// RAW output of conv2, its BatchNorm-apply + ReLU + split done by the four producer waves (no residual, no fp32 copy: those belong to
// the block boundary and stay on the 128x128 kernel).  Thread (r8, kq) of producer wave pw holds channels 4*kq .. 4*kq+3 of the rows
// pw*64 + 8*i + r8, i < 8, of the 256 x 32 tile; three register sets: a load has two K-tile periods to arrive.  Slot protocol = the
// on-the-fly form of gemm_bf3_persist_ws_kernel (iteration g: L(g+1), D(g+1) done -> transform slot g+1 -> barrier g -> D(g+3),
// L(g+4)); the weight tiles stream by LDS-DMA as before.
constexpr int kWs256BnTab = 512;      // channels of the on-the-fly operand of the 256x128 kernel (scale | shift table in LDS)
// ILV (round 4; f16x2 only, where a second set of B fragment registers fits the 168-register budget): one fragment read in the gap behind
// each matrix instruction, both operands double-buffered by k-step (see conv3x3_bf3_halo_kernel); bit-identical to ILV = 0.
template <int AK, int FMT = 0, bool BNA = false, int ABL = 0, int ILV = 0>      // ABL (measurement only, wrong results): 1 = no DMA in the loop, 2 = no tile stores, 4 = no fragment reads
__global__ void __launch_bounds__(768) gemm_bf3_persist_ws256_kernel(const Bf3Params p) {
  static_assert(ILV == 0 || FMT == 1, "interleaved fragment reads of the 256x128 kernel: f16x2 only");
  static_assert(!BNA || (FMT == 1 && AK == OPK_ROWK), "on-the-fly operand of the 256x128 kernel: f16x2, row-major");
  constexpr int BM = 256, BN = 128;
  constexpr int NPL = Bf3Fmt<FMT>::NPL;
  constexpr int NST = NPL == 2 ? 3 : 2;                            // ring stages: 3 x 48 KB (two planes per operand) or 2 x 72 KB
  constexpr int APLANE = BM * BK3, BPLANE = BN * BK3, AOPER = NPL * APLANE, BOPER = NPL * BPLANE, STAGE = AOPER + BOPER;
  constexpr int NDMA = NPL * (BM / 64) + NPL * (BN / 64);         // 18 | 12 DMA instructions per producer wave and K tile
  __shared__ __align__(1024) unsigned short smem[NST * STAGE];     // 144 KB
  __shared__ float bn_tab256[BNA ? 2 * kWs256BnTab : 1];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nkt = (p.K + BK3 - 1) / BK3;
  const int T = p.mtiles * p.ntiles, G = gridDim.x;
  const int ntl = (T - (int)blockIdx.x + G - 1) / G;
  const int total = ntl * nkt;
  if constexpr (BNA) {     // scale / shift of every input channel -> LDS (all twelve waves; one barrier, once per launch)
    for (int k = tid; k < p.K; k += 768) { bn_tab256[k] = p.a_scale[k]; bn_tab256[kWs256BnTab + k] = p.a_shift[k]; }
    __syncthreads();
  }

  if constexpr (BNA) {
  if (wave >= 8) {
    // ---------------- producer waves, on-the-fly A operand
    constexpr int NR = 8, DA = 3, NB = 2 * NPL;      // rows per thread, register sets, weight-DMA instructions per wave and slot
    const int pt = tid - 512, prow0 = (pt >> 6) * 64 + ((pt & 63) >> 3), kq = pt & 7;
    typename Bf3LoaderFor<OPK_ROWK, BN, NPL>::type lbld;
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    struct ASet { f32x4_ ra[NR]; unsigned tab, flg; };      // flg: bits 0..7 row i inside the matrix, bit 8 ragged tile (wave-uniform)
    ASet sets[DA];
    // slot iterators: slot s = (tile s / nkt of this workgroup, K tile s % nkt)
    int bj = 0, bkt = 0, lj = 0, lkt = 0;
    auto tile_id = [&](int j) { return xcd_remap(blockIdx.x + j * G, T); };
    { const int t = tile_id(0); lbld.init(p.B, (t % p.ntiles) * BN, p.N, p.K); }
    auto issue_b = [&](unsigned short* stage) {
      lbld.issue(bkt * BK3, stage + AOPER);
      if (++bkt == nkt) { bkt = 0; if (++bj < ntl) { const int t = tile_id(bj); lbld.init(p.B, (t % p.ntiles) * BN, p.N, p.K); } }
    };
    const unsigned ldb = (unsigned)p.a_ld * 4u;
    unsigned lrow[NR], lflg = 0;
    auto load_tile = [&]() {
      const int t = tile_id(lj), tm = t / p.ntiles;
      const int gr = tm * BM + prow0;
      lflg = (tm * BM + BM > p.M) ? 256u : 0u;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        lflg |= (gr + 8 * i < p.M) ? (1u << i) : 0u;
        lrow[i] = (unsigned)min(gr + 8 * i, p.M - 1) * ldb + (unsigned)kq * 16u;
      }
    };
    load_tile();
    auto load_a = [&](ASet& S) {
      const unsigned kb = (unsigned)lkt * (BK3 * 4u);
      S.tab = kb + (unsigned)kq * 16u; S.flg = lflg;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const unsigned vo = lrow[i] + kb;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(S.ra[i]) : "v"(vo), "s"(p.a_raw) : "memory");
      }
      if (++lkt == nkt) { lkt = 0; if (++lj < ntl) load_tile(); }
    };
    unsigned doff[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const unsigned r = (unsigned)(prow0 + 8 * i);
      doff[i] = r * 64u + ((((unsigned)kq >> 1) ^ ((r >> 2) & 3u)) << 4) + ((unsigned)kq & 1u) * 8u;
    }
    const float relu_floor = p.a_relu ? 0.f : -__builtin_inff();
    auto transform = [&](ASet& S, unsigned short* stage) {
      u32x4 scq, shq;
      const unsigned tab = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)bn_tab256 + S.tab;
      bf3_lds_read(scq, tab); bf3_lds_read(shq, tab + (unsigned)kWs256BnTab * 4u);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(scq), "+v"(shq)::"memory");
      const float4 s4 = __builtin_bit_cast(float4, scq), t4 = __builtin_bit_cast(float4, shq);
      const unsigned sbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)stage;
      typedef float f32x2_ __attribute__((ext_vector_type(2)));
      typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
      typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
      const f32x2_ s01 = {s4.x, s4.y}, s23 = {s4.z, s4.w}, t01 = {t4.x, t4.y}, t23 = {t4.z, t4.w};
      const bool ragged = __builtin_amdgcn_readfirstlane(S.flg) & 256u;
      float mx = 0.f;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const f32x4_ x = S.ra[i];
        const f32x2_ a01 = f32x2_{x.x, x.y} * s01 + t01, a23 = f32x2_{x.z, x.w} * s23 + t23;      // packed fp32 fma
        float v[4] = {fmaxf(a01.x, relu_floor), fmaxf(a01.y, relu_floor), fmaxf(a23.x, relu_floor), fmaxf(a23.y, relu_floor)};
        mx = fmaxf(fmaxf(fmaxf(mx, fabsf(v[0])), fmaxf(fabsf(v[1]), fabsf(v[2]))), fabsf(v[3]));
        if (ragged && !((S.flg >> i) & 1u)) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }      // rows past the end of the matrix are zero
        u32x2_ q1, q2;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const f32x2_ xx = f32x2_{v[2 * u], v[2 * u + 1]} * kF16ActScale;
          const f16x2_ h1 = __builtin_convertvector(xx, f16x2_);
          const f32x2_ r1 = xx - __builtin_convertvector(h1, f32x2_);
          q1[u] = __builtin_bit_cast(unsigned, h1); q2[u] = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, f16x2_));
        }
        const unsigned dst = sbase + doff[i];
        asm volatile("ds_write_b64 %0, %1" ::"v"(dst), "v"(q1) : "memory");
        asm volatile("ds_write_b64 %0, %1 offset:%c2" ::"v"(dst), "v"(q2), "i"(APLANE * 2) : "memory");
      }
      if (!(mx <= kF16Max / kF16ActScale)) f16x2_raise(p.status, 4u);      // overflow guard (common.h): NaN included
    };
    // ---- prologue: B slots 0..2 and A slots 0..DA-1 in flight; slot 0 transformed; then A slot DA
#pragma unroll
    for (int s0 = 0; s0 < NST; ++s0)
      if (s0 < total) issue_b(smem + s0 * STAGE);
#pragma unroll
    for (int s0 = 0; s0 < DA; ++s0)
      if (s0 < total) load_a(sets[s0]);
    const bool steady_ok = total >= 2 * DA + 2;
#define DIC_PIN_SLOT(S_)                                                                                                          \
  do { _Pragma("unroll") for (int i_ = 0; i_ < NR; ++i_) asm volatile("" : "+v"((S_).ra[i_]) :: "memory"); } while (0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DIC_PIN_SLOT(sets[0]);
    transform(sets[0], smem);
    if (DA < total) load_a(sets[0]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                  // slot 0 is in LDS
    int st = 0;
    for (int g0 = 0; g0 < total; g0 += DA) {
#pragma unroll
      for (int u = 0; u < DA; ++u) {
        const int g = g0 + u;
        if (g >= total) break;
        const int stn = st == NST - 1 ? 0 : st + 1;
        if (g + 1 < total) {
          // D(g+1) and L(g+1) must have landed; younger in issue order: L(g+2), D(g+2), L(g+3) (see the header)
          if (steady_ok && g >= 2 && g + DA < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NR + NB) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          DIC_PIN_SLOT(sets[(u + 1) % DA]);
          transform(sets[(u + 1) % DA], smem + stn * STAGE);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                              // slot g+1 is in LDS; the consumers are done with stage st
        if (g + NST < total) issue_b(smem + st * STAGE);
        if (g + 1 + DA < total) load_a(sets[(u + 1) % DA]);
        st = stn;
      }
    }
#undef DIC_PIN_SLOT
    return;
  }
  } else
  if (wave >= 8) {
    // ---------------- producer waves (8..11 -> row groups 0..3 of the loaders)
    typename Bf3LoaderFor<AK, BM, NPL>::type la;
    typename Bf3LoaderFor<OPK_ROWK, BN, NPL>::type lbld;
    int pj = 0, pkt = 0;
    {
      const int t = xcd_remap(blockIdx.x, T);
      la.init(p.A, (t / p.ntiles) * BM, p.M, p.K);
      lbld.init(p.B, (t % p.ntiles) * BN, p.N, p.K);
    }
    auto prefetch = [&](unsigned short* stage) {
      la.issue(pkt * BK3, stage);
      lbld.issue(pkt * BK3, stage + AOPER);
      if (++pkt == nkt) {
        pkt = 0; ++pj;
        if (pj < ntl) {
          const int t = xcd_remap(blockIdx.x + pj * G, T);
          la.init(p.A, (t / p.ntiles) * BM, p.M, p.K);
          lbld.init(p.B, (t % p.ntiles) * BN, p.N, p.K);
        }
      }
    };
#pragma unroll
    for (int s0 = 0; s0 < NST; ++s0)
      if (s0 < total) prefetch(smem + s0 * STAGE);
    if (total >= NST) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * (NST - 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                  // slot 0 is in LDS
    int st = 0;
    for (int g = 0; g < total; ++g) {
      // slot g+1 must have landed before the consumers read it (after this barrier); with three stages slot g+2 may stay in flight
      if (NST >= 3 && g + 2 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                // ... and the consumers are done with stage st
      if (!(ABL & 1) && g + NST < total) prefetch(smem + st * STAGE);
      st = st == NST - 1 ? 0 : st + 1;
    }
    return;
  }

  // ---------------- consumer waves
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int i31 = lane & 31, h = lane >> 5, key = (i31 >> 2) & 3;
  const unsigned offA = (unsigned)((wm * 64 + i31) * 64), offB = (unsigned)((wn * 64 + i31) * 64);
  const unsigned pos[2] = {(unsigned)(((0 + h) ^ key) * 16), (unsigned)(((2 + h) ^ key) * 16)};
  const unsigned sbase0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)smem;
  u32x4 fa[2][2][3], fb[2][3];                     // A: [k-step buffer][tile][plane];  B: [tile][plane]
#define DIC_W_READ_A(KS_, SB_)                                                                                       \
  if constexpr (!(ABL & 4)) _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl)                   \
      bf3_lds_read(fa[KS_][i][pl], (SB_) + (unsigned)(pl * APLANE * 2) + offA + (unsigned)(i * 32 * 64) + pos[KS_]);
#define DIC_W_READ_B(KS_, SB_)                                                                                       \
  if constexpr (!(ABL & 4)) _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl)                   \
      bf3_lds_read(fb[j][pl], (SB_) + (unsigned)(AOPER * 2 + pl * BPLANE * 2) + offB + (unsigned)(j * 32 * 64) + pos[KS_]);
#define DIC_W_PIN(KS_)                                                                                               \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl) {                 \
    asm volatile("" : "+v"(fa[KS_][i][pl])); asm volatile("" : "+v"(fb[i][pl])); }
#define DIC_W_MFMA(KS_, PA_, PB_)                                                                                    \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                        \
      acc[i][j] = bf3_mfma<FMT>(fa[KS_][i][PA_], fb[j][PB_], acc[i][j]);
#define DIC_W_MFMA_ALL(KS_)                                                                                          \
  if constexpr (NPL == 3) { DIC_W_MFMA(KS_, 2, 0) DIC_W_MFMA(KS_, 0, 2) DIC_W_MFMA(KS_, 1, 1) }                     \
  DIC_W_MFMA(KS_, 1, 0) DIC_W_MFMA(KS_, 0, 1) DIC_W_MFMA(KS_, 0, 0)
  __builtin_amdgcn_s_barrier();                                    // slot 0 is in LDS
  u32x4 ga[2][2][ILV ? 2 : 1], gb[2][2][ILV ? 2 : 1];              // ILV form: [k-step buffer][tile][plane] for both operands
  if constexpr (ILV != 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        bf3_lds_read(ga[0][i][pl], sbase0 + (unsigned)(pl * APLANE * 2) + offA + (unsigned)(i * 32 * 64) + pos[0]);
        bf3_lds_read(gb[0][i][pl], sbase0 + (unsigned)(AOPER * 2 + pl * BPLANE * 2) + offB + (unsigned)(i * 32 * 64) + pos[0]);
      }
  } else {
    DIC_W_READ_A(0, sbase0)
  }
  int g = 0, st = 0;
  for (int j = 0; j < ntl; ++j) {
    for (int kt = 0; kt < nkt; ++kt, ++g) {
      const int stn = st == NST - 1 ? 0 : st + 1;
      const unsigned sb = sbase0 + (unsigned)(st * STAGE) * 2u, sbn = sbase0 + (unsigned)(stn * STAGE) * 2u;
      st = stn;
      if constexpr (ILV != 0) {
#define DIC_W_PIN2(KS_)                                                                                              \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) {                   \
    asm volatile("" : "+v"(ga[KS_][i][pl])); asm volatile("" : "+v"(gb[KS_][i][pl])); }
        // ---- k-step 0 (fragments requested during the previous slot's second half); k-step 1's fragments in the gaps
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        DIC_W_PIN2(0)
        __builtin_amdgcn_sched_barrier(0);
        const unsigned a1 = sb + offA + pos[1], b1 = sb + (unsigned)(AOPER * 2) + offB + pos[1];
        bf3_static_for<0, 12>([&](auto mc) {
          constexpr int m = decltype(mc)::value, ai = (m & 3) >> 1, aj = m & 1;
          acc[ai][aj] = bf3_mfma<FMT>(ga[0][ai][bf3_prod_plane_a(2, m >> 2)], gb[0][aj][bf3_prod_plane_b(2, m >> 2)], acc[ai][aj]);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (m < 4) {
            if constexpr (!(ABL & 4)) bf3_lds_read_off<(m % 2) * APLANE * 2 + (m / 2) * 32 * 64>(ga[1][m / 2][m % 2], a1);
            __builtin_amdgcn_sched_barrier(0);
          } else if constexpr (m < 8) {
            if constexpr (!(ABL & 4)) bf3_lds_read_off<(m % 2) * BPLANE * 2 + ((m - 4) / 2) * 32 * 64>(gb[1][(m - 4) / 2][m % 2], b1);
            __builtin_amdgcn_sched_barrier(0);
          }
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        DIC_W_PIN2(1)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- k-step 1; k-step 0 of the next slot in the gaps (past the last slot: a harmless re-read of a ring stage)
        const unsigned a0 = sbn + offA + pos[0], b0 = sbn + (unsigned)(AOPER * 2) + offB + pos[0];
        bf3_static_for<0, 12>([&](auto mc) {
          constexpr int m = decltype(mc)::value, ai = (m & 3) >> 1, aj = m & 1;
          acc[ai][aj] = bf3_mfma<FMT>(ga[1][ai][bf3_prod_plane_a(2, m >> 2)], gb[1][aj][bf3_prod_plane_b(2, m >> 2)], acc[ai][aj]);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (m < 4) {
            if constexpr (!(ABL & 4)) bf3_lds_read_off<(m % 2) * APLANE * 2 + (m / 2) * 32 * 64>(ga[0][m / 2][m % 2], a0);
            __builtin_amdgcn_sched_barrier(0);
          } else if constexpr (m < 8) {
            if constexpr (!(ABL & 4)) bf3_lds_read_off<(m % 2) * BPLANE * 2 + ((m - 4) / 2) * 32 * 64>(gb[0][(m - 4) / 2][m % 2], b0);
            __builtin_amdgcn_sched_barrier(0);
          }
        });
#undef DIC_W_PIN2
        continue;
      }
      // k-step 0: its A fragments were requested one k-step ago; request its B fragments and k-step 1's A fragments
      DIC_W_READ_B(0, sb) DIC_W_READ_A(1, sb)
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * NPL) : "memory");
      DIC_W_PIN(0)
      __builtin_amdgcn_sched_barrier(0);
      DIC_W_MFMA_ALL(0)
      __builtin_amdgcn_sched_barrier(0);
      // k-step 1: B fragments (same registers: the MFMAs above have been issued), then every read of this stage is done
      DIC_W_READ_B(1, sb)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      DIC_W_PIN(1)
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NPL == 3) { DIC_W_MFMA(1, 2, 0) } else { DIC_W_MFMA(1, 1, 0) }
      if (g + 1 < total) { DIC_W_READ_A(0, sbn) }
      if constexpr (NPL == 3) { DIC_W_MFMA(1, 0, 2) DIC_W_MFMA(1, 1, 1) DIC_W_MFMA(1, 1, 0) }
      DIC_W_MFMA(1, 0, 1) DIC_W_MFMA(1, 0, 0)
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- seam: store the tile
    const int t = xcd_remap(blockIdx.x + j * G, T);
    const int tm = t / p.ntiles, tn = t - tm * p.ntiles;
    const bool full = (tm + 1) * BM <= p.M && (tn + 1) * BN <= p.N;
    const int n0 = tn * BN + wn * 64 + (lane & 31), m0 = tm * BM + wm * 64 + 4 * h;
    float cs[2] = {0.f, 0.f}, cs2[2] = {0.f, 0.f};
    if constexpr (FMT == 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) acc[i][jj] *= ep_alpha(p.ep);
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float* col = p.ep.C + (long long)(m0 + i * 32) * p.ep.ldc + n0 + jj * 32;
        if constexpr ((ABL & 2) != 0) {
          if (acc[i][jj][0] == 1.2345e-30f) col[0] = 0.f;      // (keeps the accumulators alive)
        } else
        if (full) {
#pragma unroll
          for (int r = 0; r < 16; ++r) col[(long long)((r & 3) + 8 * (r >> 2)) * p.ep.ldc] = acc[i][jj][r];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (m0 + i * 32 + (r & 3) + 8 * (r >> 2) < p.M && n0 + jj * 32 < p.N)
              col[(long long)((r & 3) + 8 * (r >> 2)) * p.ep.ldc] = acc[i][jj][r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {       // rows beyond M hold exact zeros (zero-filled operand rows)
          cs[jj] += acc[i][jj][r]; cs2[jj] += acc[i][jj][r] * acc[i][jj][r];
          acc[i][jj][r] = 0.f;
        }
      }
    if (p.ep.stats) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const float a = cs[jj] + __shfl_xor(cs[jj], 32, 64), b = cs2[jj] + __shfl_xor(cs2[jj], 32, 64);
        const int n = n0 + jj * 32;
        if (lane < 32 && n < p.N && tm * BM + wm * 64 < p.M) {      // (wave tiles entirely past M have no row in the table)
          p.ep.stats[((long long)(tm * 4 + wm) * 2 + 0) * p.N + n] = a;
          p.ep.stats[((long long)(tm * 4 + wm) * 2 + 1) * p.N + n] = b;
        }
      }
    }
  }
#undef DIC_W_READ_A
#undef DIC_W_READ_B
#undef DIC_W_PIN
#undef DIC_W_MFMA
#undef DIC_W_MFMA_ALL
}

// 3x3 / stride-1 / pad-1 convolution on 14x14 maps (ResNet layer 3: 36 of the 50 3x3 convolutions) with the input tile's HALO
// staged in LDS instead of an im2col gather.  Why: the contraction kernels of this file are bound by operand intake per CU
// (see gemm_bf3_persist_kernel), and the gather is the worst customer - every tap re-fetches the same pixels as 64-B halves
// of 128-B lines, 24 KB per K tile.  Here the K loop is ordered (32-channel chunk, tap): the 13 padded image rows x 16
// padded pixels that the 128 output pixels of a tile touch are copied ONCE per chunk (39 KB incl. zero padding, whole
// 128-B lines), and the nine taps read their A fragments from that image at shifted pixel offsets; only the weights
// (24 KB per tap) still stream: 24 + 39/9 = 28 KB per K tile instead of 48.
//   * padded-row index space: image b owns rows b*(H+1)+1 .. b*(H+1)+H, the rows b*(H+1) are zero and shared between
//     neighbouring images; padded column 0 and W+1.. are zero; a tile's rows are pr_lo .. pr_lo+12 (<= 13 for any tile);
//   * LDS: 2 halo buffers (chunk parity) x 3 planes x 13 rows x 16 pixels x 64 B + a 3-stage ring of weight tiles + 1 KB that
//     absorbs one dummy DMA per chunk so that every wave issues exactly 10 halo instructions (counted vmcnt);
//   * pipeline = gemm_bf3_persist_kernel: persistent over output tiles, fragments double-buffered by k-step, one barrier
//     per K tile, weights three tiles ahead, the next chunk's halo issued at tap 0 of the current chunk (8 K tiles early);
//   * summation order per output: chunk-major, tap-minor (the other kernels: tap-major) - same products, fp32-level
//     differences in the last bit, not bit-identical to them.
//   * BNA (round 4, f16x2 only): the input is the RAW fp32 output of the convolution before (p.a_raw, [M][C]) and the halo image is
//     act(raw * scale[c] + shift[c]) formed by the producer waves - the BatchNorm-apply + ReLU + split pass that used to write the
//     planes (bn_apply_planes: 8 B per element of HBM traffic and a launch) is gone.  Once per 32-channel chunk, i.e. once per nine
//     K tiles: thread (pixel slot, channel quad) loads 7 x 16 B of the next chunk at tap 0 (complete by tap 2: the counted waits
//     for the weight tiles issued after them cover them), transforms and writes the two plane images at tap 2, six K tiles before
//     the computing waves first read them.  Padding pixels are written as zeros (the padding applies to the activation).
constexpr int kHaloBnTab = 512;       // channels of the on-the-fly operand of the halo kernel (scale | shift table in LDS)
//     58 / 7 for 56x56 maps (ResNet layer 1, 64 -> 64 channels: 2 x 52 KB of halo images + 48 KB of weight ring; its 64 output channels
//     take the left half of the 128-column tile - weight rows 64..127 are out of the buffer's range and load as zeros, the epilogue's
//     column guard drops them).  PARKED (experiments build, switch 127): correct (scripts/experiments/test_parked_kernels_gpu.py) and
//     no faster - 110 us per launch against 17 + 85 us for the planes pass + gathered kernel it replaces (seven tiles of two chunks per
//     workgroup: a 13-slot transform, a tile seam and a halo set-up per 18 K tiles), pipelined step 8.61 ms with it, 8.59 without;
//   * HROW / RMAX: padded pixels per halo row and halo rows of a tile - 16 / 13 for 14x14 maps; 32 / 9 for 28x28 maps (ResNet layer 2;
//     BNA form only: its halo buffers hold the two f16x2 planes, 2 x 36 KB, where three planes of that size would not fit).  RMAX is the
//     largest number of padded rows any 128-pixel tile touches (brute force over every tile start: 13 | 9).
//   * ILV (round 4): the computing waves issue ONE fragment read (and its address arithmetic) in the gap behind each matrix instruction
//     instead of eight reads in a row in front of twelve matrix instructions: a wave issues in order, so a block of reads + ~20 address
//     instructions in front of the first MFMA of a k-step is ~150 cycles during which the matrix pipe of its SIMD (one computing wave per
//     SIMD) has nothing to do, while one ds_read_b128 + 2-3 vector instructions fit the 32-cycle shadow of an MFMA.  Same products, same
//     order per accumulator: bit-identical to ILV = 0.
template <int ABL, int FMT = 0, bool BNA = false, int HROW = 16, int RMAX = 13, int ILV = 0>      // FMT: operand format; ABL (measurement only): 1 = no weight DMA in the loop, 2 = no halo DMA in the loop, 3 = neither
__global__ void __launch_bounds__(512) conv3x3_bf3_halo_kernel(const Bf3Params p) {
  static_assert(!BNA || FMT == 1, "on-the-fly halo operand: f16x2 only");
  static_assert((HROW == 16 && RMAX == 13) || (BNA && HROW == 32 && RMAX == 9) || (BNA && HROW == 58 && RMAX == 7),
                "halo geometry: 14x14 maps, or 28x28 / 56x56 with the on-the-fly operand");
  constexpr int BM = 128, BN = 128, NSTB = 3;
  constexpr int HPLANE = RMAX * HROW * BK3, HBUF = (BNA ? Bf3Fmt<FMT>::NPL : 3) * HPLANE;        // elements: 13 KB per plane, 39 KB per buffer (14x14)
  constexpr int BPLANE = BN * BK3, BSTAGE = (BNA ? Bf3Fmt<FMT>::NPL : 3) * BPLANE;               // 24 KB per weight tile (16 KB in the on-the-fly form: two planes)
  constexpr int NPL = Bf3Fmt<FMT>::NPL;                               // planes in use (buffers keep three plane slots)
  constexpr int NHALO = (RMAX * NPL + 3) / 4;                          // halo DMA instructions per producer wave and chunk (10 | 7)
  constexpr int NB = 2 * NPL;                                          // weight DMA instructions per producer wave and K tile
  __shared__ __align__(1024) unsigned short smem[2 * HBUF + NSTB * BSTAGE + 512];
  unsigned short* const bring = smem + 2 * HBUF;
  unsigned short* const dummy = smem + 2 * HBUF + NSTB * BSTAGE;
  __shared__ float halo_bn_tab[BNA ? 2 * kHaloBnTab : 1];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.A.g.H, W = p.A.g.W, C = p.A.g.C, ohw = H * W, nimg = p.M / ohw;
  const int NC = C / BK3;
  const int T = p.mtiles * p.ntiles, G = gridDim.x;
  // Work list as in gemm_bf3_persist_ws_kernel: whole tiles, then (remainder-round K split, tail_split > 1) at most one piece =
  // the channel chunks [pc0, pc0 + npc) of remainder tile F + blockIdx / tail_split, all nine taps each.
  const bool split = p.tail_split > 1;
  const int F = split ? p.tail_first_tile : T;
  const int ntl = (F - (int)blockIdx.x + G - 1) / G;
  const int perc = split ? (NC + p.tail_split - 1) / p.tail_split : 0;
  const bool has_piece = split && (int)blockIdx.x < (T - F) * p.tail_split;
  const int piece_tile = has_piece ? F + (int)blockIdx.x / p.tail_split : 0;
  const int pc0 = has_piece ? ((int)blockIdx.x % p.tail_split) * perc : 0;
  const int npc = has_piece ? min(NC, pc0 + perc) - pc0 : 0;
  const int nwork = ntl + (has_piece ? 1 : 0);
  const int nchunks = ntl * NC + npc, total = nchunks * 9;
  if (total == 0) return;
  if constexpr (BNA) {     // scale / shift of every input channel -> LDS (all eight waves; one barrier, once per launch)
    for (int k = tid; k < C; k += 512) { halo_bn_tab[k] = p.a_scale[k]; halo_bn_tab[kHaloBnTab + k] = p.a_shift[k]; }
    __syncthreads();
  }

  auto tile_of = [&](int j, int& tm, int& tn) {
    const int t = j < ntl ? xcd_remap(blockIdx.x + j * G, F) : piece_tile;
    tm = t / p.ntiles; tn = t - tm * p.ntiles;
  };
  auto row_lo = [&](int tm) {                     // first padded row of the tile's halo
    const int m0 = tm * BM, b0 = m0 / ohw, oy0 = (m0 - b0 * ohw) / W;
    return b0 * (H + 1) + oy0;
  };

  if (wave >= 4) {
    // ================= producer waves (4..7 -> w = 0..3): all LDS-DMA of the workgroup
    const int w = wave - 4;
    // halo chunk n = (tile n / NC, channels 32*(n % NC) ..) into buffer n & 1.  The source offsets of this wave's ten
    // (row, plane) instructions depend on the tile only: derived once per tile (integer divisions), reused by its NC chunks
    int hoff[NHALO];                                        // element offset of this lane's 16 bytes, chunk 0
    unsigned hok = 0u;                                      // bit t: instruction t reads real data (else the zero line)
    auto setup_halo = [&](int j) {
      int tm, tn;
      tile_of(j, tm, tn);
      const int pr_lo = row_lo(tm), px = lane >> 2, ix = px - 1;
      int b = pr_lo / (H + 1), rr = pr_lo - b * (H + 1);    // padded row pr_lo + r = image b, row rr (0 = the shared zero row)
      int r_prev = 0;
      hok = 0u;
#pragma unroll
      for (int t = 0; t < NHALO; ++t) {
        const int idx = w + 4 * t;                          // (row, plane) = (idx / NPL, idx % NPL); rows >= 13 = the dummies
        const int r = NPL == 3 ? (idx * 43) >> 7 : idx >> 1;
        for (int a = r_prev; a < r; ++a) { if (++rr == H + 1) { rr = 0; ++b; } }      // at most two steps
        r_prev = r;
        const int q = r * HROW + px, c16 = (lane & 3) ^ ((q >> 2) & 3);
        const int pix = (b * H + rr - 1) * W + ix;
        hoff[t] = (pix >> 1) * (C * 2) + ((pix & 1) << 5) + c16 * 8;
        if (r < RMAX && rr != 0 && b < nimg && (unsigned)ix < (unsigned)W) hok |= 1u << t;
      }
    };
    auto issue_halo = [&](int n) {                          // n-th chunk of the work list
      const bool in_piece = n >= ntl * NC;
      const int cc = in_piece ? pc0 + (n - ntl * NC) : n % NC;
      if (in_piece ? n == ntl * NC : cc == 0) setup_halo(in_piece ? ntl : n / NC);
      unsigned short* buf = smem + (n & 1) * HBUF;
#pragma unroll
      for (int t = 0; t < NHALO; ++t) {
        const int idx = w + 4 * t, r = NPL == 3 ? (idx * 43) >> 7 : idx >> 1, pl = idx - NPL * r;
        const unsigned short* src = ((hok >> t) & 1u) ? p.A.p[pl] + (hoff[t] + cc * 64) : g_zero_line16;
        unsigned short* dst = r < RMAX ? buf + pl * HPLANE + r * (HROW * BK3) : dummy;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    };
    // ---- on-the-fly operand (BNA): thread (ps, l8) of the 256 producer threads owns channels 4*l8 .. 4*l8+3 of the pixel slots
    // q = ps + 32*i, i < NRAW (14x14: 13 rows x 16 padded pixels = 208 slots, NRAW = 7, i = 6 exists for ps < 16 only; 28x28: 9 x 32 =
    // 288 slots, NRAW = 9).  Every thread ALWAYS issues its NRAW loads (slots that are padding, beyond the batch or beyond the buffer
    // read element 0 and are replaced by zeros): the counted waits rely on the instruction count.
    constexpr int NRAW = (RMAX * HROW + 31) / 32;
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    f32x4_ ra[BNA ? NRAW : 1];
#pragma unroll
    for (int i = 0; i < (BNA ? NRAW : 1); ++i) ra[i] = f32x4_{0.f, 0.f, 0.f, 0.f};
    unsigned roff[BNA ? NRAW : 1], rdst[BNA ? NRAW : 1];
    unsigned rok = 0u, rcc = 0u;                             // bit i: slot i holds a real pixel; channel chunk of the loads in flight
    const int pt = tid - 256, l8 = pt & 7, ps = pt >> 3;
    if constexpr (BNA) {
#pragma unroll
      for (int i = 0; i < NRAW; ++i) {
        const unsigned q = (unsigned)(ps + 32 * i);
        rdst[i] = q * 64u + ((((unsigned)l8 >> 1) ^ ((q >> 2) & 3u)) << 4) + ((unsigned)l8 & 1u) * 8u;
      }
    }
    auto setup_raw = [&](int j) {
      int tm, tn;
      tile_of(j, tm, tn);
      const int pr_lo = row_lo(tm);
      // (one division per tile, wave-uniform; a slot's padded row pr_lo + r, r < RMAX <= H + 1, crosses at most one image boundary)
      const int b_lo = __builtin_amdgcn_readfirstlane(pr_lo / (H + 1)), rr_lo = __builtin_amdgcn_readfirstlane(pr_lo - b_lo * (H + 1));
      rok = 0u;
#pragma unroll
      for (int i = 0; i < NRAW; ++i) {
        const int q = ps + 32 * i, r = q / HROW, ix = (q % HROW) - 1;
        const bool wrap = rr_lo + r >= H + 1;
        const int b = b_lo + (wrap ? 1 : 0), rr = rr_lo + r - (wrap ? H + 1 : 0);
        const bool ok = r < RMAX && rr != 0 && b < nimg && (unsigned)ix < (unsigned)W;
        const int pix = ok ? (b * H + rr - 1) * W + ix : 0;
        roff[i] = (unsigned)pix * ((unsigned)p.a_ld * 4u) + (unsigned)l8 * 16u;
        if (ok) rok |= 1u << i;
      }
    };
    auto issue_raw = [&](int n) {                            // n-th chunk of the work list: its loads into `ra`
      const bool in_piece = n >= ntl * NC;
      const int cc = in_piece ? pc0 + (n - ntl * NC) : n % NC;
      if (in_piece ? n == ntl * NC : cc == 0) setup_raw(in_piece ? ntl : n / NC);
      rcc = (unsigned)cc;
#pragma unroll
      for (int i = 0; i < NRAW; ++i) {
        const unsigned vo = roff[i] + (unsigned)cc * 128u;
        // ("+v": the destination IS the register the value lives in across the tap loop - a fresh output register would have to be
        //  copied into the loop-carried one right here, before the data has arrived)
        asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(ra[i]) : "v"(vo), "s"(p.a_raw) : "memory");
      }
    };
    const float relu_floor = p.a_relu ? 0.f : -__builtin_inff();
    auto transform_raw = [&](int n) {                        // registers -> the two plane images of halo buffer n & 1 (after the loads have landed)
#pragma unroll
      for (int i = 0; i < NRAW; ++i) asm volatile("" : "+v"(ra[i]) :: "memory");
      u32x4 scq, shq;
      const unsigned tab = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)halo_bn_tab + (rcc * 32u + (unsigned)l8 * 4u) * 4u;
      bf3_lds_read(scq, tab); bf3_lds_read(shq, tab + (unsigned)kHaloBnTab * 4u);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(scq), "+v"(shq)::"memory");
      const float4 s4 = __builtin_bit_cast(float4, scq), t4 = __builtin_bit_cast(float4, shq);
      const unsigned hbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)(smem + (n & 1) * HBUF);
      typedef float f32x2_ __attribute__((ext_vector_type(2)));
      typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
      typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
      float mx = 0.f;
#pragma unroll
      for (int i = 0; i < NRAW; ++i) {
        if (i == NRAW - 1 && ps >= (RMAX * HROW - 32 * (NRAW - 1))) break;      // slot beyond the buffer (14x14: 208 slots; wave-uniform: ps = 8 * wave + lane / 8)
        const bool ok = (rok >> i) & 1u;
        float v[4];
        v[0] = ok ? fmaxf(fmaf(ra[i].x, s4.x, t4.x), relu_floor) : 0.f;
        v[1] = ok ? fmaxf(fmaf(ra[i].y, s4.y, t4.y), relu_floor) : 0.f;
        v[2] = ok ? fmaxf(fmaf(ra[i].z, s4.z, t4.z), relu_floor) : 0.f;
        v[3] = ok ? fmaxf(fmaf(ra[i].w, s4.w, t4.w), relu_floor) : 0.f;
        mx = fmaxf(fmaxf(fmaxf(mx, fabsf(v[0])), fmaxf(fabsf(v[1]), fabsf(v[2]))), fabsf(v[3]));
        u32x2_ q1, q2;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const f32x2_ xx = f32x2_{v[2 * u], v[2 * u + 1]} * kF16ActScale;
          const f16x2_ h1 = __builtin_convertvector(xx, f16x2_);
          const f32x2_ r1 = xx - __builtin_convertvector(h1, f32x2_);
          q1[u] = __builtin_bit_cast(unsigned, h1); q2[u] = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, f16x2_));
        }
        const unsigned dst = hbase + rdst[i];
        asm volatile("ds_write_b64 %0, %1" ::"v"(dst), "v"(q1) : "memory");
        asm volatile("ds_write_b64 %0, %1 offset:%c2" ::"v"(dst), "v"(q2), "i"(HPLANE * 2) : "memory");
      }
      // overflow guard (common.h): a NaN input survives neither fmaxf nor this compare - !(mx <= bound) catches it
      if (!(mx <= kF16Max / kF16ActScale)) f16x2_raise(p.status, 4u);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // weight slot s = (tile, chunk, tap) in that order, K offset tap*C + 32*chunk
    typename Bf3LoaderFor<OPK_ROWK, BN, NPL>::type lbld;
    int pj = 0, pcc = ntl > 0 ? 0 : pc0, pend = ntl > 0 ? NC : pc0 + npc, ptap = 0, pst = 0;
    { int tm, tn; tile_of(0, tm, tn); lbld.init(p.B, tn * BN, p.N, p.K); }
    auto issue_b = [&]() {
      lbld.issue(ptap * C + pcc * BK3, bring + pst * BSTAGE);
      pst = pst == NSTB - 1 ? 0 : pst + 1;
      if (++ptap == 9) {
        ptap = 0;
        if (++pcc == pend) {
          ++pj;
          pcc = pj < ntl ? 0 : pc0; pend = pj < ntl ? NC : pc0 + npc;
          if (pj < nwork) { int tm, tn; tile_of(pj, tm, tn); lbld.init(p.B, tn * BN, p.N, p.K); }
        }
      }
    };
    constexpr int NHI = BNA ? NRAW : NHALO;        // vector-memory instructions a halo chunk costs this wave
    if constexpr (BNA) issue_raw(0); else issue_halo(0);
#pragma unroll
    for (int s0 = 0; s0 < NSTB; ++s0)
      if (s0 < total) issue_b();
    if (total >= NSTB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NB) : "memory");     // halo chunk 0 and weight tile 0 (NB each younger tile)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (BNA) transform_raw(0);
    __builtin_amdgcn_s_barrier();
    int g = 0, n = 0;
    bool halo_prev = false;                        // the previous slot issued a halo chunk (10 instructions before its weights)
    for (int j = 0; j < nwork; ++j)
      for (int cc = 0; cc < (j < ntl ? NC : npc); ++cc, ++n)
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap, ++g) {
          // weight tile g+1 (and, before tap 0 of a chunk, that chunk's halo - older still) must have landed; younger, in issue
          // order: the previous slot's halo chunk (10, if it issued one) and weight tile g+2 (6)
          if (g + 2 >= total) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          else if (halo_prev) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NHI + NB) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB) : "memory");
          __builtin_amdgcn_s_barrier();
          halo_prev = false;
          if (tap == 0 && n + 1 < nchunks) {
            if constexpr (BNA) { if (!(ABL & 2)) issue_raw(n + 1); } else if (!(ABL & 2)) issue_halo(n + 1);
            halo_prev = !(ABL & 2);
          }
          if (!(ABL & 1) && g + NSTB < total) issue_b();
          // (BNA) the loads issued at tap 0 are older than weight tile g+3 of that tap, which the wait of tap 2 has seen land
          if constexpr (BNA) if (!(ABL & 2) && tap == 2 && n + 1 < nchunks) transform_raw(n + 1);
        }
    return;
  }

  // ================= consumer waves
  const int wm = wave >> 1, wn = wave & 1;
  const int i31 = lane & 31, h = lane >> 5;
  auto pixel_base = [&](int tm, int i) {                // LDS pixel index of the top-left tap of this lane's output row
    int m = tm * BM + wm * 64 + i * 32 + i31;
    m = min(m, p.M - 1);
    const int b = m / ohw, rem = m - b * ohw, oy = rem / W, ox = rem - oy * W;
    return (b * (H + 1) + oy - row_lo(tm)) * HROW + ox;
  };
  const unsigned sbase0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)smem;
  const unsigned offB = (unsigned)((wn * 64 + i31) * 64), bkey = (unsigned)((i31 >> 2) & 3);
  const unsigned posB[2] = {(unsigned)(((0 + h) ^ bkey) * 16), (unsigned)(((2 + h) ^ bkey) * 16)};

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  u32x4 fa[2][2][3], fb[2][2][3];
  // A fragments of k-step KS_: pixel q = QB_[i] + 16*kh + kw of halo buffer HB_, 16-B slot (2*KS_ + h) ^ key(q)
#define DIC_HALO_READ_A(KS_, HB_, QB_, TAPOFF_)                                                                      \
  if constexpr (!(ABL & 8)) _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                    \
    const unsigned q = (unsigned)((QB_)[i] + (TAPOFF_));                                                             \
    const unsigned a = sbase0 + (unsigned)((HB_) * HBUF * 2) + q * 64u + ((((unsigned)(2 * (KS_) + h)) ^ ((q >> 2) & 3u)) << 4); \
    _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl) bf3_lds_read(fa[KS_][i][pl], a + (unsigned)(pl * HPLANE * 2)); \
  }
#define DIC_HALO_READ_B(KS_, ST_)                                                                                    \
  if constexpr (!(ABL & 4)) _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl)                   \
      bf3_lds_read(fb[KS_][j][pl], sbase0 + (unsigned)((2 * HBUF + (ST_) * BSTAGE + pl * BPLANE) * 2) + offB + (unsigned)(j * 32 * 64) + posB[KS_]);
#define DIC_PIPE_PIN(KS_)                                                                                            \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl) {                 \
    asm volatile("" : "+v"(fa[KS_][i][pl])); asm volatile("" : "+v"(fb[KS_][i][pl])); }
#define DIC_PIPE_MFMA(KS_, PA_, PB_)                                                                                 \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                        \
      acc[i][j] = bf3_mfma<FMT>(fa[KS_][i][PA_], fb[KS_][j][PB_], acc[i][j]);

  int qb[2], qbn[2];
  { int tm, tn; tile_of(0, tm, tn); qb[0] = pixel_base(tm, 0); qb[1] = pixel_base(tm, 1); }
  qbn[0] = qb[0]; qbn[1] = qb[1];
  __builtin_amdgcn_s_barrier();                    // halo chunk 0 and weight tile 0 are in LDS
  // ---- ILV form: addresses kept in registers.  A fragment of k-step 0: aaddr[i] (plane pl at +pl*HPLANE*2 as an instruction offset);
  // k-step 1 = the same address with bit 5 flipped ((2 + h) ^ key = (h ^ key) ^ 2); B fragments: stage base + bl[k-step]
  unsigned aaddr[2] = {0u, 0u};
  const unsigned bl[2] = {offB + posB[0], offB + posB[1]};
  auto a_addr0 = [&](int hbuf, int q) {
    return sbase0 + (unsigned)(hbuf * HBUF * 2) + (unsigned)q * 64u + ((((unsigned)h) ^ (((unsigned)q >> 2) & 3u)) << 4);
  };
  if constexpr (ILV == 0) {
    DIC_HALO_READ_A(0, 0, qb, 0) DIC_HALO_READ_B(0, 0)
  } else {
    aaddr[0] = a_addr0(0, qb[0]); aaddr[1] = a_addr0(0, qb[1]);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) bf3_lds_read(fa[0][i][pl], aaddr[i] + (unsigned)(pl * HPLANE * 2));
    DIC_HALO_READ_B(0, 0)
  }
  // MFMA m of a k-step: product m / 4 of the format's list, accumulator ((m % 4) / 2, m % 2); read r of a k-step: r < 2 NPL = A fragment
  // (tile r / NPL, plane r % NPL), else B fragment
  constexpr int NP = NPL == 3 ? 6 : 3;

  int g = 0, st = 0, n = 0;                        // slot, its weight stage, its (global) chunk
  for (int j = 0; j < nwork; ++j) {
    const int ncj = j < ntl ? NC : npc;            // chunks of this work item (the piece: a slice of the channels)
    for (int cc = 0; cc < ncj; ++cc, ++n) {
      const int hb = n & 1;
#pragma unroll 1
      for (int tap = 0; tap < 9; ++tap, ++g) {
        const int stn = st == NSTB - 1 ? 0 : st + 1;
        const int kh = tap >= 6 ? 2 : tap >= 3 ? 1 : 0, kw = tap - 3 * kh, tapoff = kh * HROW + kw;
        if constexpr (ILV != 0) {
          // ---- k-step 0 (fragments requested during the previous slot's second half); k-step 1's fragments in the gaps
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          DIC_PIPE_PIN(0)
          __builtin_amdgcn_sched_barrier(0);
          // (the next slot - next tap / next chunk = other buffer / next tile; past the last slot: a harmless re-read - is chosen here and its
          //  A addresses are formed in two of this k-step's read-free gaps)
          const int t1 = tap + 1, kh1 = t1 >= 6 ? 2 : t1 >= 3 ? 1 : 0;
          const bool same_chunk = tap < 8, next_tile = !same_chunk && cc + 1 >= ncj;
          const int nhb = same_chunk ? hb : hb ^ 1, noff = same_chunk ? kh1 * HROW + (t1 - 3 * kh1) : 0;
          unsigned baddr = 0u, an[2] = {0u, 0u};
          bf3_static_for<0, 4 * NP>([&](auto mc) {
            constexpr int m = decltype(mc)::value, ai = (m & 3) >> 1, aj = m & 1;
            acc[ai][aj] = bf3_mfma<FMT>(fa[0][ai][bf3_prod_plane_a(NPL, m >> 2)], fb[0][aj][bf3_prod_plane_b(NPL, m >> 2)], acc[ai][aj]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (m < 2 * NPL) {
              if constexpr (!(ABL & 8)) bf3_lds_read_off<(m % NPL) * HPLANE * 2>(fa[1][m / NPL][m % NPL], aaddr[m / NPL] ^ 32u);
              __builtin_amdgcn_sched_barrier(0);
            } else if constexpr (m < 4 * NPL) {
              if constexpr (m == 2 * NPL) baddr = sbase0 + (unsigned)((2 * HBUF + st * BSTAGE) * 2) + bl[1];
              if constexpr (!(ABL & 4)) bf3_lds_read_off<(m % NPL) * BPLANE * 2 + ((m - 2 * NPL) / NPL) * 32 * 64>(fb[1][(m - 2 * NPL) / NPL][m % NPL], baddr);
              __builtin_amdgcn_sched_barrier(0);
            } else if constexpr (m < 4 * NPL + 2) {
              an[m - 4 * NPL] = a_addr0(nhb, (next_tile ? qbn[m - 4 * NPL] : qb[m - 4 * NPL]) + noff);
              __builtin_amdgcn_sched_barrier(0);
            }
          });
          aaddr[0] = an[0]; aaddr[1] = an[1];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          DIC_PIPE_PIN(1)
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          // ---- k-step 1; k-step 0 of the NEXT slot in the gaps
          bf3_static_for<0, 4 * NP>([&](auto mc) {
            constexpr int m = decltype(mc)::value, ai = (m & 3) >> 1, aj = m & 1;
            acc[ai][aj] = bf3_mfma<FMT>(fa[1][ai][bf3_prod_plane_a(NPL, m >> 2)], fb[1][aj][bf3_prod_plane_b(NPL, m >> 2)], acc[ai][aj]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (m < 2 * NPL) {
              if constexpr (!(ABL & 8)) bf3_lds_read_off<(m % NPL) * HPLANE * 2>(fa[0][m / NPL][m % NPL], aaddr[m / NPL]);
              __builtin_amdgcn_sched_barrier(0);
            } else if constexpr (m < 4 * NPL) {
              if constexpr (m == 2 * NPL) baddr = sbase0 + (unsigned)((2 * HBUF + stn * BSTAGE) * 2) + bl[0];
              if constexpr (!(ABL & 4)) bf3_lds_read_off<(m % NPL) * BPLANE * 2 + ((m - 2 * NPL) / NPL) * 32 * 64>(fb[0][(m - 2 * NPL) / NPL][m % NPL], baddr);
              __builtin_amdgcn_sched_barrier(0);
            }
          });
          if (tap == 1 && cc == ncj - 1 && j + 1 < nwork) {     // next tile's pixel bases, well before its first fragment reads
            int tm, tn; tile_of(j + 1, tm, tn);
            qbn[0] = pixel_base(tm, 0); qbn[1] = pixel_base(tm, 1);
          }
          st = stn;
          continue;
        }
        DIC_HALO_READ_A(1, hb, qb, tapoff) DIC_HALO_READ_B(1, st)
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(4 * NPL) : "memory");
        DIC_PIPE_PIN(0)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NPL == 3) { DIC_PIPE_MFMA(0, 2, 0) DIC_PIPE_MFMA(0, 0, 2) DIC_PIPE_MFMA(0, 1, 1) }
        DIC_PIPE_MFMA(0, 1, 0) DIC_PIPE_MFMA(0, 0, 1) DIC_PIPE_MFMA(0, 0, 0)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        DIC_PIPE_PIN(1)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NPL == 3) { DIC_PIPE_MFMA(1, 2, 0) } else { DIC_PIPE_MFMA(1, 1, 0) }
        if (g + 1 < total) {                       // k-step 0 of the next slot: next tap / next chunk (other buffer) / next tile
          if (tap < 8) {
            const int t1 = tap + 1, kh1 = t1 >= 6 ? 2 : t1 >= 3 ? 1 : 0, off1 = kh1 * HROW + (t1 - 3 * kh1);
            DIC_HALO_READ_A(0, hb, qb, off1)
          } else if (cc + 1 < ncj) {
            DIC_HALO_READ_A(0, hb ^ 1, qb, 0)
          } else {
            DIC_HALO_READ_A(0, hb ^ 1, qbn, 0)
          }
          DIC_HALO_READ_B(0, stn)
        }
        if constexpr (NPL == 3) { DIC_PIPE_MFMA(1, 0, 2) DIC_PIPE_MFMA(1, 1, 1) DIC_PIPE_MFMA(1, 1, 0) }
        if (tap == 1 && cc == ncj - 1 && j + 1 < nwork) {     // next tile's pixel bases, well before its first fragment reads
          int tm, tn; tile_of(j + 1, tm, tn);
          qbn[0] = pixel_base(tm, 0); qbn[1] = pixel_base(tm, 1);
        }
        DIC_PIPE_MFMA(1, 0, 1) DIC_PIPE_MFMA(1, 0, 0)
        __builtin_amdgcn_sched_barrier(0);
        st = stn;
      }
    }
    // ---- seam (the piece of a remainder tile: same stores into this wave's [64][64] slab of raw partial sums in tail_ws;
    // rows past M hold junk there, the fix-up masks them)
    const bool piece = j >= ntl;
    int tm, tn;
    tile_of(j, tm, tn);
    const bool full = piece || ((tm + 1) * BM <= p.M && (tn + 1) * BN <= p.N);
    const int n0 = piece ? (lane & 31) : tn * BN + wn * 64 + (lane & 31), m0 = piece ? 4 * h : tm * BM + wm * 64 + 4 * h;
    float* const Cb = piece ? p.tail_ws + ((long long)(((int)blockIdx.x / p.tail_split) * 4 + wave) * p.tail_split + (int)blockIdx.x % p.tail_split) * (64 * 64)
                            : p.ep.C;
    const long long ldc = piece ? 64 : p.ep.ldc;
    float cs[2] = {0.f, 0.f}, cs2[2] = {0.f, 0.f};
    if constexpr (FMT == 1) {             // undo the operands' power-of-two scales (a piece stays raw: the fix-up scales the sum)
      if (!piece) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) acc[i][jj] *= ep_alpha(p.ep);
      }
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float* col = Cb + (long long)(m0 + i * 32) * ldc + n0 + jj * 32;
        if (full) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            col[(long long)((r & 3) + 8 * (r >> 2)) * ldc] = acc[i][jj][r];
            cs[jj] += acc[i][jj][r]; cs2[jj] += acc[i][jj][r] * acc[i][jj][r];
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (m0 + i * 32 + (r & 3) + 8 * (r >> 2) < p.M && n0 + jj * 32 < p.N) {   // rows past M repeat the last pixel: masked
              col[(long long)((r & 3) + 8 * (r >> 2)) * ldc] = acc[i][jj][r];
              cs[jj] += acc[i][jj][r]; cs2[jj] += acc[i][jj][r] * acc[i][jj][r];
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
      }
    if (p.ep.stats && !piece) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const float a = cs[jj] + __shfl_xor(cs[jj], 32, 64), b = cs2[jj] + __shfl_xor(cs2[jj], 32, 64);
        const int nn = n0 + jj * 32;
        if (lane < 32 && nn < p.N) {
          p.ep.stats[((long long)(tm * 2 + wm) * 2 + 0) * p.N + nn] = a;
          p.ep.stats[((long long)(tm * 2 + wm) * 2 + 1) * p.N + nn] = b;
        }
      }
    }
    qb[0] = qbn[0]; qb[1] = qbn[1];
  }
#undef DIC_HALO_READ_A
#undef DIC_HALO_READ_B
#undef DIC_PIPE_PIN
#undef DIC_PIPE_MFMA
}


#ifdef DIC_EXPERIMENTS
#include "experiments/conv1x1_astat.inc"      // parked: A-stationary conv3 kernel with the BatchNorm-apply fused in (see the note in the file)
#endif

}  // namespace dic

namespace dic {

static int g_last_mtiles = 0;   // M tiles of the most recent launch (row count of the BN partial-sum table)
#ifdef DIC_EXPERIMENTS
void conv1x1_astat_switch(int on);
#endif
// Kernel-selection switches (dic_debug_force_staged_gemm, include/dic.h).  The product library keeps the ones its tests use to
// compare kernels that the policy below really selects (tile forcing 11 / 21 / 24 / 20, persistent policy 70 / 73 / 79, halo
// 74 / 75 / 78); the ablations and the parked kernels exist only in the experiments build (-DDIC_EXPERIMENTS).
static int g_bf3_force = 0;            // 11 / 21 force the 64x64 / 128x64 workgroup tile, 24 the persistent 128x128 kernel wherever its epilogue applies
static int g_bf3_persist_grid = 224;   // persistent kernels: at most this many workgroups (one per CU).  224 rather than 256: same time per launch
                                       // (operand delivery, not CU count, bounds them) and the main stream's short kernels find free CUs: pipelined step
                                       // 14.28 -> 14.02 ms
static int g_bf3_halo_ilv = 1;         // codes 120 / 121: computing waves of the f16x2 128x128 kernels (LDS-halo on-the-fly form, persistent 1x1 / gathered) read their fragments in a block / one per MFMA gap (default)
static int g_bf3_halo = 1;             // 3x3 convolutions of 14x14 maps on the LDS-halo kernel: 0 = off (code 75), 1 = from 128 tiles (78, default), 2 = always (74)
static int g_bf3_persist_policy = 4;   // codes 70..73, 79: 0 = never, 1 = only K <= 64, 2 = also K <= 256 on >= 3072-tile grids, 3 = 1x1 convolutions by CU fill, 4 = also the gathered (im2col) ones
static int g_bf3_tail_mode = 0;        // codes 60..63: 1 = no remainder-tile K split, 2 = split also for T >= 7*256, 3 = split by 4 at most
static int g_bf3_remainder_split = 1;  // persistent kernels: remainder-round K split on (default) / off (codes 91 / 90)
static int g_bf3_narrow_bn = 1;        // on-the-fly-operand 1x1 kernel for 64 output channels (codes 96 / 97)
static int g_bf3_halo28 = 1;           // 3x3 convolutions of 28x28 maps with the on-the-fly operand on the LDS-halo kernel (codes 94 / 95)
static int g_bf3_few_remap = 1;        // few-tiles launches: slice z on XCD z (codes 92 / 93)
static int g_bf3_wgrad_persist = 1;    // weight gradients with 32..255 output tiles of 128x128: on the persistent kernel, every tile in K slices (codes 118 / 119)
static int g_bf3_remainder_grid = 256; // ... and the workgroups such a launch may use
static int g_bf3_ws256 = 1;            // codes 80 / 81: 256x128 form of the warp-specialised kernel for f16x2 row-major plain-epilogue launches by policy (default) / never
static int g_bf3_slots = 4;            // codes 114 / 115: input slots in flight per producer wave of the eight-producer form: 4 (default) / 6
                                       // (eight do not fit 168 registers: the compiler spills, and a spilled destination of an in-flight
                                       //  asm load is garbage - that build hung the kernel; build.py now refuses any spilling kernel)
static int g_bf3_producers = 8;        // codes 112 / 113: producer waves of the f16x2 on-the-fly-operand kernel: 4 / 8 (default)
#ifdef DIC_EXPERIMENTS
static int g_bf3_halo56 = 0;           // parked: 3x3 convolutions of 56x56 maps (64 or n x 128 output channels) with the on-the-fly operand on the LDS-halo kernel (codes 126 / 127)
static int g_bf3_ws256_bn = 0;         // parked: on-the-fly operand (no residual, no copy) on the 256x128 kernel (codes 106 / 107)
static int g_bf3_stages = 2;           // ring depth of the 128-wide variants (42 / 43)
static int g_bf3_ws = 1;               // codes 76 / 77: persistent kernel in its warp-specialised form on / off
static int g_bf3_ablate = 0;           // 1 = no DMA in the loop, 2 = also no LDS fragment reads (64x64 rowk only)
static int g_bf3_bn_ablate = 0;        // on-the-fly-operand kernel, timing only (wrong results): bit 0 = no residual read, bit 1 = no fp32 copy written (57..59)
static int g_bf3_ws256_ablate = 0;     // 256x128 kernel (f16x2, planes), timing only: 1 = no DMA in the loop, 2 = no tile stores, 3 = neither, 4 = no fragment reads (44..47)
static int g_bf3_halo_bna_ablate = 0;  // LDS-halo kernel, on-the-fly form, timing only (wrong results): bit 0 = no weight DMA in the loop, bit 1 = no input loads / transform in the loop (64..66)
#else
constexpr int g_bf3_stages = 2, g_bf3_ws = 1, g_bf3_ablate = 0, g_bf3_halo56 = 0;
#endif
template <int AK, int TM, int TN>
static void launch_bf3_variant(const Bf3Params& p, int blocks, hipStream_t st) {
#ifdef DIC_EXPERIMENTS
  if constexpr (TM == 2) {
    if (g_bf3_stages == 3) { hipLaunchKernelGGL((gemm_bf3_kernel<AK, TM, TN, 3>), dim3(blocks), dim3(256), 0, st, p); return; }
  }
#endif
  if (p.fmt == 1) { hipLaunchKernelGGL((gemm_bf3_kernel<AK, TM, TN, 2, 0, 1>), dim3(blocks), dim3(256), 0, st, p); return; }
  hipLaunchKernelGGL((gemm_bf3_kernel<AK, TM, TN, 2>), dim3(blocks), dim3(256), 0, st, p);
}

// Workgroups per persistent launch (dic_conv_persistent_grid, include/dic.h): a launch never asks for more than this many CUs,
// so that the convolutions of SEVERAL ResNet forwards in flight (engine.py) run side by side instead of taking turns.
int gemm_bf3_set_persist_grid(int workgroups) {
  if (workgroups < 1 || workgroups > 1024) return -1;
  g_bf3_persist_grid = workgroups;
  return 0;
}

int gemm_bf3_force_tile(int code) {      // 0 = accepted, -1 = unknown in this build
  if (code == 70 || code == 73) { g_bf3_persist_policy = code - 70; return 0; }
  if (code == 79) { g_bf3_persist_policy = 4; return 0; }
  if (code == 74 || code == 75 || code == 78) { g_bf3_halo = code == 74 ? 2 : code == 78 ? 1 : 0; return 0; }
#ifdef DIC_EXPERIMENTS
  if (code == 76) { g_bf3_ws = 1; return 0; }
#else
  if (code == 76) return 0;                                      // warp-specialised persistent kernel: the only form of the product
#endif
  if (code == 20 || code == 11 || code == 21 || code == 24) { g_bf3_force = code == 20 ? 0 : code; return 0; }
  if (code == 90 || code == 91) { g_bf3_remainder_split = code - 90; return 0; }      // remainder-round K split of the persistent kernels off / on (default)
  if (code == 80 || code == 81) { g_bf3_ws256 = code == 80; return 0; }              // 256x128 form for f16x2 1x1 convolutions: by policy (default) / never
  if (code == 112 || code == 113) { g_bf3_producers = code == 112 ? 4 : 8; return 0; }      // f16x2 on-the-fly-operand kernel: four / eight (default) producer waves
  if (code == 114 || code == 115) { g_bf3_slots = code == 114 ? 4 : 6; return 0; }          // ... its input slots in flight per producer wave: four (default) / six
  if (code == 92 || code == 93) { g_bf3_few_remap = code - 92; return 0; }                  // few-tiles launches (every tile in K slices): plain order / slice z on XCD z (default)
  if (code == 94 || code == 95) { g_bf3_halo28 = code - 94; return 0; }                      // LDS-halo kernel for 28x28 maps (on-the-fly operand form): never / by policy (default)
  if (code == 96 || code == 97) { g_bf3_narrow_bn = code - 96; return 0; }                   // on-the-fly-operand 1x1 kernel for CO = 64 (layer 1's conv1): never / by policy (default)
  if (code == 120 || code == 121) { g_bf3_halo_ilv = code - 120; return 0; }               // LDS-halo kernel, on-the-fly form: fragment reads in a block / interleaved with the MFMAs (default)
  if (code == 118 || code == 119) { g_bf3_wgrad_persist = code - 118; return 0; }          // weight gradients: 64x64 tiles with the caller's K split / persistent 128x128 kernel, every tile in K slices (default)
#ifdef DIC_EXPERIMENTS
  if (code == 126 || code == 127) { g_bf3_halo56 = code - 126; return 0; }            // parked: LDS-halo kernel for 56x56 maps (on-the-fly operand form): never (default) / by policy
  if (code == 106 || code == 107) { g_bf3_ws256_bn = code - 106; return 0; }        // parked: conv3-style on-the-fly operand (no residual, no copy) on the 256x128 kernel: never (default) / by policy
  if (code == 110 || code == 111) { conv1x1_astat_switch(code - 110); return 0; }    // parked: conv3 (K = 128 / 256, f16x2) on the A-stationary kernel: never (default) / by shape
  if (code == 71 || code == 72) { g_bf3_persist_policy = code - 70; return 0; }
  if (code >= 60 && code <= 63) { g_bf3_tail_mode = code - 60; return 0; }
  if (code >= 82 && code <= 89) { g_bf3_persist_grid = 256 - 16 * (code - 82); return 0; }      // persistent grids of at most 256, 240, ... 144 workgroups
  if (code == 42 || code == 43) { g_bf3_stages = code - 40; return 0; }
  if (code >= 50 && code <= 56) { g_bf3_ablate = code - 50; if (code == 50) g_bf3_bn_ablate = g_bf3_halo_bna_ablate = g_bf3_ws256_ablate = 0; return 0; }      // (54 / 55: persistent kernel with cache-hot A / A and B)
  if (code >= 57 && code <= 59) { g_bf3_bn_ablate = code - 56; return 0; }       // (50 clears it)
  if (code >= 44 && code <= 47) { g_bf3_ws256_ablate = code - 43; return 0; }       // (50 clears it)
  if (code >= 64 && code <= 69) { g_bf3_halo_bna_ablate = code <= 66 ? code - 63 : code == 67 ? 4 : code == 68 ? 12 : 8; return 0; }  // (50 clears it; 67 / 68 / 69: no B / no A and B / no A fragment reads)
  if (code == 77) { g_bf3_ws = 0; return 0; }
  if (code == 22 || code == 23 || code == 26) { g_bf3_force = code; return 0; }
#endif
  return -1;
}

static int launch_bf3(Bf3Params p, hipStream_t st, float* tail_ws, int splitk = 1, float* splitk_ws = nullptr,
                      const BnFuseArgs* bn_fuse = nullptr, int* bn_fused = nullptr, int tail_ws_slabs = 256, bool probe = false) {
  // tile choice (measured, scripts/bench_bf3.py): bigger per-wave tiles halve the LDS fragment traffic per MFMA and
  // amortise the per-K-tile barrier, but need >= ~2 workgroups per CU to keep 256 CUs busy
  int tmv = 1, tnv = 1;
  // Measured on MI355X (scripts/bench_bf3b.py): with 72-96 KB of LDS the 128-wide tiles run at 2 or 1 workgroup per
  // CU and lose to 64x64 (3 per CU) almost everywhere; 128x64 wins on very deep grids (4096^3: 153 vs 109 TF-eq) and,
  // optionally (policy 31), on the 3-tiles-per-CU shapes of ResNet layer 3 (M=12544, N=256).
  const long long t11 = (long long)ceil_div(p.M, 64) * ceil_div(p.N, 64);
  if (t11 >= 16384 && p.K >= 1024) { tmv = 2; tnv = 1; }
  // batch-256-scale grids (scripts/bench_bf3_b256.py): 128x64 wins by 4..16 % except on the narrow shallow shapes
  // (N <= 256 and K <= 1024); neutral at batch 64, +1.2 % on the batch-256 step
  if (t11 >= 6144 && !(p.N <= 256 && p.K <= 1024)) { tmv = 2; tnv = 1; }
  // 512..1023 tiles of 64x64 (ResNet layer 3 at batch 64: 784 = one round of 768 + a 16-tile remainder that needs the
  // K-split + fix-up): 128x64 tiles make it a single round of 392 workgroups at two per CU.  Measured on the whole
  // ResNet forward (scripts/bench_resnet_ab.py 11 20): 15.39 -> 15.18 ms.
  if (t11 >= 512 && t11 < 1024 && p.N <= 256 && p.K >= 512) { tmv = 2; tnv = 1; }
  // long-K grids of >= 1024 tiles (layer-3/4 shapes at batch 256): 128x64 wins 7..13 % per launch (scripts/bench_bf3_b256.py);
  // neutral on the batch-64 forward, -0.8 % on the batch-256 forward (scripts/bench_resnet_ab.py --batch 256)
  if (t11 >= 1024 && p.K >= 1024) { tmv = 2; tnv = 1; }
  if (g_bf3_force == 11) { tmv = 1; tnv = 1; }
  if (g_bf3_force == 21) { tmv = 2; tnv = 1; }
  if (g_bf3_force == 22 || g_bf3_force == 23) { tmv = 2; tnv = 2; }
  // Persistent 128x128 kernel: plain-store epilogue without bias, at least two K tiles.  Policy (scripts/bench_bf3_pipe.py):
  // it wins where a tile is only a few K tiles long and the grid is many rounds deep - the 1x1 expansions 64 -> 256
  // (-12 % per launch at batch 64, -20 % at batch 256), and at batch-256 scale also 128 -> 512 and 256 -> 1024.
  const bool plain_ep = !p.ep.bias && !p.ep.accumulate && p.ep.act == ACT_NONE;      // what the halo kernel (and the parked forms) store
  // (row-major operands of the persistent kernels go through 32-bit buffer offsets: K in whole 32-element tiles, planes below 2 GiB)
  const bool buf_ok = p.K % BK3 == 0 && p.B.ld % 32 == 0 && (long long)(p.N + 1) * p.B.ld * 2 < (1ll << 31) &&
                      (p.A.kind != OPK_ROWK || (p.A.ld % 32 == 0 && (long long)(p.M + 1) * p.A.ld * 2 < (1ll << 31)));
  // (64 output channels - ResNet layer 1's conv1 - only for the on-the-fly operand: the weight rows 64..127 of the 128-column tile are
  //  out of the buffer's range and load as zeros, the epilogue's column guard drops them; twice the matrix work of a kernel that runs
  //  at a seventh of the matrix pipe, in exchange for the 820-MB pass that would otherwise write that block output as planes)
  const bool narrow_bn = g_bf3_narrow_bn != 0 && p.a_raw && p.N == 64 && p.A.kind == OPK_ROWK && p.fmt == 1;
  // (64 output channels on the LDS-halo kernel for 56x56 maps - layer 1's conv2 - in the same way)
  const bool halo56 = g_bf3_halo56 != 0 && p.A.kind == OPK_IM2COL && p.A.g.H == 56 && p.A.g.W == 56 && p.a_raw && p.fmt == 1 && !p.a_res && !p.a_out &&
                      p.A.g.C <= kHaloBnTab && (p.N == 64 || p.N % 128 == 0);
  const bool narrow = narrow_bn || (halo56 && p.N == 64);
  const bool persist_ok = splitk <= 1 && !p.ep.row_map && !p.ep.C2 && p.K > BK3 && (p.N % 128 == 0 || narrow) && (plain_ep || g_bf3_ws) && buf_ok;
  const long long t22 = (long long)ceil_div(p.M, 128) * ceil_div(p.N, 128);
  const int rounds22 = (int)((t22 + g_bf3_persist_grid - 1) / g_bf3_persist_grid);
  double fill22 = (double)t22 / ((double)rounds22 * g_bf3_persist_grid);      // how evenly the tiles divide among the CUs
  {   // ... or, with the remainder-round K split (below), among all of them
    const int gmax = g_bf3_remainder_grid, r = (int)(t22 % gmax), fullr = (int)(t22 / gmax), units = ceil_div(p.K, BK3);
    if (g_bf3_remainder_split && tail_ws && splitk <= 1 && fullr >= 1 && r > 0 && units >= 4) {
      int sp = std::min(std::min(std::min(gmax / r, units / 2), 16), tail_ws_slabs / 4 / r);
      while (sp > 1 && (sp - 1) * ceil_div(units, sp) >= units) --sp;
      if (sp >= 2) fill22 = std::max(fill22, (double)t22 / ((fullr + (double)ceil_div(units, sp) / units + 0.15) * gmax));
    }
  }
  // Few tiles, long K (ResNet layer 4 at batch 64: 100 tiles of K = 2048, or K = 4608 gathered): fewer tiles than CUs, each a long
  // dependent K loop.  The remainder-round machinery with NO whole round - every tile cut into `few_sp` K slices of at least 8 K
  // tiles, one slice per workgroup, summed by the tail fix-up - puts 2-5x as many CUs on the launch (103 -> ~45 us at those shapes).
  int few_sp = 0;
  {
    const int gmax = g_bf3_remainder_grid, units = ceil_div(p.K, BK3);
    if (g_bf3_remainder_split && tail_ws && splitk <= 1 && t22 >= 32 && t22 < gmax && units >= 32 && !narrow_bn) {
      int sp = std::min(std::min(std::min(gmax / (int)t22, units / 8), 16), tail_ws_slabs / 4 / (int)t22);
      while (sp > 1 && (sp - 1) * ceil_div(units, sp) >= units) --sp;
      if (sp >= 2 && t22 * sp >= 160) few_sp = sp;
    }
  }
  bool persist = false;
  if (persist_ok && g_bf3_force == 0) {
    if (g_bf3_persist_policy == 1) persist = p.K <= 64 && t22 >= 1024;
    else if (g_bf3_persist_policy == 2) persist = (p.K <= 64 && t22 >= 1024) || (p.K <= 256 && t22 >= 3072);
    else if (g_bf3_persist_policy >= 3)     // warp-specialised form (scripts/bench_bf3_pipe.py at batch 64 and 256): wins wherever the
      persist = ((p.A.kind == OPK_ROWK || g_bf3_persist_policy >= 4) && t22 >= 192 && (fill22 >= 0.85 || (fill22 >= 0.75 && p.K >= 512))) ||      // tiles fill the CUs
                (p.K <= 64 && t22 >= 1024) || (p.K <= 256 && t22 >= 3072) ||
                ((p.A.kind == OPK_ROWK || g_bf3_persist_policy >= 4) && few_sp > 0);
  }
  if (g_bf3_force == 24 || g_bf3_force == 26) persist = persist_ok;
  bool ws256 = false;
  {   // 256x128 form (two computing waves per SIMD): half as many tiles must still fill the CUs.  f16x2 row-major launches with the
      // plain epilogue (the ResNet's 1x1 expansions and downsample convolutions): measured 15-19 % faster there, slower below ~190 tiles
    const long long t42 = (long long)ceil_div(p.M, 256) * ceil_div(p.N, 128);
    const int rounds42 = (int)((t42 + g_bf3_persist_grid - 1) / g_bf3_persist_grid);
    const double fill42 = (double)t42 / ((double)rounds42 * g_bf3_persist_grid);
    // (with the on-the-fly operand - round 4, switch 107: conv3 reading conv2's raw output - only without residual / fp32 copy)
#ifdef DIC_EXPERIMENTS      // parked (see the kernel's header): correct, bit-identical to the plane route, neutral in the step
    const bool bna_ok = !p.a_raw || (g_bf3_ws256_bn != 0 && !p.a_res && !p.a_out && p.K >= 128 && p.K <= kWs256BnTab);
#else
    const bool bna_ok = !p.a_raw;
#endif
    if (g_bf3_force == 0 && g_bf3_ws256 != 0 && p.fmt == 1 && bna_ok && p.A.kind == OPK_ROWK && persist && few_sp == 0 && g_bf3_ws && plain_ep &&
        t42 >= 192 && fill42 >= 0.75 && p.K >= 64)
      ws256 = true;
#ifdef DIC_EXPERIMENTS
    if (g_bf3_force == 26) ws256 = persist_ok && plain_ep;
#endif
  }
  // 3x3 convolutions of 14x14 maps: the LDS-halo kernel
  const ConvGeom& cg = p.A.g;
  // (28x28 maps - ResNet layer 2 - only in the form that takes the raw input and forms the activation in the producer waves, f16x2)
  const bool halo28 = g_bf3_halo28 != 0 && cg.H == 28 && cg.W == 28 && p.a_raw && p.fmt == 1 && !p.a_res && !p.a_out && cg.C <= kHaloBnTab;
  const bool halo = g_bf3_halo != 0 && g_bf3_force == 0 && persist_ok && plain_ep && (t22 >= 128 || g_bf3_halo == 2) && p.A.kind == OPK_IM2COL && p.A.paired && cg.KH == 3 && cg.KW == 3 &&
                    cg.stride == 1 && cg.pad == 1 && ((cg.H == 14 && cg.W == 14) || halo28 || halo56) && cg.nchw == 0 && cg.C % BK3 == 0 &&
                    p.M % (cg.H * cg.W) == 0 && p.K == 9 * cg.C;
  if (halo) persist = true;
  if (persist) { tmv = 2; tnv = 2; }
  const bool pipe = tmv == 2 && tnv == 2 && g_bf3_force != 22;       // 128x128 (experiments build: the deep-pipelined kernel when not persistent; 22: the plain loop)
  persist = persist && pipe;
  p.mtiles = ceil_div(p.M, 64 * tmv); p.ntiles = ceil_div(p.N, 64 * tnv);
  g_last_mtiles = p.mtiles;
  p.splitk = 1; p.ws = nullptr;
  if (p.fmt != 1) p.ep.alpha = 1.0f;
  const int T = p.mtiles * p.ntiles, nk = ceil_div(p.K, BK3);
  int total = T;
  p.tail_first_block = T; p.tail_first_tile = 0; p.tail_split = 1; p.tail_ws = nullptr;
  int tail_tiles = 0;
  if (splitk > 1) {   // split-K over every tile (weight gradients: few tiles, very long K): all tiles go through the
                      // slice + tail_fixup machinery, slices summed in fixed order
    DIC_REQUIRE(splitk_ws != nullptr && splitk <= 16, "gemm_bf3: split-K needs a workspace and <= 16 slices");
    tmv = 1; tnv = 1;
    p.mtiles = ceil_div(p.M, 64); p.ntiles = ceil_div(p.N, 64);
    g_last_mtiles = p.mtiles;
    const int T2 = p.mtiles * p.ntiles;
    splitk = std::min(splitk, std::max(1, nk / 2));
    tail_tiles = T2; p.tail_first_tile = 0; p.tail_first_block = 0; p.tail_split = splitk; p.tail_ws = splitk_ws;
    total = T2 * splitk;
  } else
  if (tail_ws && tmv == 1 && tnv == 1 && g_bf3_tail_mode != 1) {
    const int r = T % 256;
    int sp = r > 0 ? 256 / r : 0;
    sp = std::min(sp, std::min(nk / 2, g_bf3_tail_mode == 3 ? 4 : 16));
    if (r > 0 && r <= 128 && sp >= 2 && (T < 7 * 256 || g_bf3_tail_mode == 2)) {
      tail_tiles = r; p.tail_first_tile = T - r; p.tail_first_block = T - r; p.tail_split = sp; p.tail_ws = tail_ws;
      total = (T - r) + r * sp;
    }
  }
  const bool im = p.A.kind == OPK_IM2COL;
  // Remainder-round K split of the persistent 128x128 kernels: T tiles on G workgroups leave T mod G tiles for a last, partly
  // filled round.  Those r tiles are cut into `sp` K slices (1x1 / gathered: K tiles; halo: channel chunks) handed to the first
  // r * sp workgroups as one extra piece each, and finished by the tail fix-up (slice sums in K order + epilogue + BatchNorm
  // partials; deterministic).  Taken when it shortens the longest workgroup by at least a fifth of a tile.
  int persist_grid = 0, rem128 = 0;
  if (persist && !ws256 && g_bf3_ablate == 0 && (halo || g_bf3_ws)) {
    persist_grid = ceil_div(T, ceil_div(T, g_bf3_persist_grid));
    const int units = halo ? cg.C / BK3 : nk;                 // what a slice is made of
    const int gmax = g_bf3_remainder_grid;                     // CUs a split launch may use
    const int r = T % gmax, fullr = T / gmax;
    if (fullr == 0 && few_sp > 0 && !halo && g_bf3_remainder_split && tail_ws) {      // few tiles, long K: every tile in slices
      persist_grid = r * few_sp; rem128 = r;
      p.tail_first_tile = 0; p.tail_split = few_sp; p.tail_ws = tail_ws;
      p.few_remap = g_bf3_few_remap;
    } else
    if (g_bf3_remainder_split && tail_ws && (plain_ep || !halo) && fullr >= 1 && r > 0 && units >= 4 && !narrow) {      // (the fix-up works on whole 64x64 quadrants of N % 128 == 0)
      // tail_ws holds tail_ws_slabs slabs of [64][64] floats (kGemmTailWsBytes = 256 for callers of the C ABI, 1024 inside the
      // ResNet workspace); a piece writes four (one per consumer wave): r * sp <= slabs / 4
      const int kSlabs = tail_ws_slabs;
      int sp = std::min(std::min(std::min(gmax / r, units / 2), 16), kSlabs / 4 / r);
      while (sp > 1 && (sp - 1) * ceil_div(units, sp) >= units) --sp;      // no empty slice
      const double longest_now = (double)ceil_div(T, persist_grid);
      const double longest_split = fullr + (sp > 1 ? (double)ceil_div(units, sp) / units : 1.0) + 0.15;     // + the fix-up launch
      if (sp >= 2 && r * sp <= gmax && r * 4 * sp <= kSlabs && longest_split + 0.2 <= longest_now) {
        persist_grid = gmax; rem128 = r;
        p.tail_first_tile = T - r; p.tail_split = sp; p.tail_ws = tail_ws;
      }
    }
  }
  const bool halo_bna = halo && p.a_raw && p.fmt == 1 && !p.a_res && !p.a_out && cg.C <= kHaloBnTab && g_bf3_ablate == 0;      // 3x3 halo kernel with the on-the-fly operand
  const bool ws256_bna = ws256 && persist && !halo && p.a_raw != nullptr;
  if (p.a_raw && !halo_bna && !ws256_bna && !(persist && !halo && !ws256 && g_bf3_ws && g_bf3_ablate == 0 && !im)) {      // on-the-fly operand: persistent 1x1 kernel, the halo kernel, or nothing
    if (!probe && !im)      // (a caller that gets 1 takes the plane route; one that cannot - dic_debug_conv1x1_bn* - reports this text)
      set_last_error("conv1x1 with on-the-fly BatchNorm operand: shape M=%d C=%d -> CO=%d is not eligible (the launch policy keeps it off "
                     "the persistent 128x128 kernel: needs CO %% 128 == 0, C %% 32 == 0, C > 32 and enough output tiles to fill the CUs); "
                     "nothing was launched", p.M, p.K, p.N);
    return 1;
  }
  if (probe) return DIC_OK;                 // conv1x1_bf3_bn_eligible: the decision only
  // algorithmic HBM bytes of the launch (bench.py roofline): every operand element read once, the output written once.  Plane operands
  // cost 2 B per plane and element; an im2col A operand is its input image (each pixel read once, not once per tap); the on-the-fly
  // operand reads the raw fp32 tensor (+ the residual) and may write the fp32 copy of its input
  double abytes = (double)p.N * p.K * 2.0 * (p.fmt == 1 ? 2 : 3) + (double)p.M * p.N * 4.0;
  if (p.a_raw && im) abytes += (double)(p.M / std::max(1, cg.OH * cg.OW)) * cg.H * cg.W * cg.C * 4.0;      // (halo kernel: the raw input image once)
  else if (p.a_raw) abytes += (double)p.M * p.K * 4.0 * (1 + (p.a_res ? 1 : 0) + (p.a_out ? 1 : 0));
  else if (im) abytes += (double)(p.M / std::max(1, cg.OH * cg.OW)) * cg.H * cg.W * cg.C * 2.0 * (p.fmt == 1 ? 2 : 3);
  else abytes += (double)p.M * p.K * 2.0 * (p.fmt == 1 ? 2 : 3);
  gemm_profile_mark_begin(st, 2.0 * p.M * p.N * (double)p.K, (p.fmt == 1 ? 3000 : 2000) + (p.a_raw ? OPK_ROWK_BN : p.A.kind) * 10 + (halo ? 6 : (persist && ws256) ? 7 : (persist && !g_bf3_ws) ? 8 : persist ? 5 : pipe ? 4 : (tmv - 1) * 2 + (tnv - 1)), abytes);
  if (persist && (halo || !ws256) && (halo || g_bf3_ws) && g_bf3_ablate == 0) {      // the product's 128x128 kernels
    g_last_mtiles = 2 * p.mtiles;          // statistics rows per 64-row wave tile
    // as few workgroups as give the same number of tiles per workgroup: the CUs left over serve the other stream's kernels
    const int grid = persist_grid;
#ifdef DIC_EXPERIMENTS
    if (p.a_raw && (g_bf3_bn_ablate & 1)) p.a_res = nullptr;
    if (p.a_raw && (g_bf3_bn_ablate & 2)) p.a_out = nullptr;
#endif
    if (p.fmt == 1) {
#ifdef DIC_EXPERIMENTS
      if (halo_bna && cg.W == 56) hipLaunchKernelGGL((conv3x3_bf3_halo_kernel<0, 1, true, 58, 7, 1>), dim3(grid), dim3(512), 0, st, p);
      else
#endif
      if (halo_bna && cg.W == 28 && g_bf3_halo_ilv) hipLaunchKernelGGL((conv3x3_bf3_halo_kernel<0, 1, true, 32, 9, 1>), dim3(grid), dim3(512), 0, st, p);
      else if (halo_bna && cg.W == 28) hipLaunchKernelGGL((conv3x3_bf3_halo_kernel<0, 1, true, 32, 9>), dim3(grid), dim3(512), 0, st, p);
#ifdef DIC_EXPERIMENTS
      else if (halo_bna && g_bf3_halo_bna_ablate == 1) hipLaunchKernelGGL((conv3x3_bf3_halo_kernel<1, 1, true, 16, 13, 1>), dim3(grid), dim3(512), 0, st, p);
      else if (halo_bna && g_bf3_halo_bna_ablate == 2) hipLaunchKernelGGL((conv3x3_bf3_halo_kernel<2, 1, true, 16, 13, 1>), dim3(grid), dim3(512), 0, st, p);
      else if (halo_bna && g_bf3_halo_bna_ablate == 3) hipLaunchKernelGGL((conv3x3_bf3_halo_kernel<3, 1, true, 16, 13, 1>), dim3(grid), dim3(512), 0, st, p);
      else if (halo_bna && g_bf3_halo_bna_ablate == 4) hipLaunchKernelGGL((conv3x3_bf3_halo_kernel<4, 1, true, 16, 13, 1>), dim3(grid), dim3(512), 0, st, p);
      else if (halo_bna && g_bf3_halo_bna_ablate == 8) hipLaunchKernelGGL((conv3x3_bf3_halo_kernel<8, 1, true, 16, 13, 1>), dim3(grid), dim3(512), 0, st, p);
      else if (halo_bna && g_bf3_halo_bna_ablate == 12) hipLaunchKernelGGL((conv3x3_bf3_halo_kernel<12, 1, true, 16, 13, 1>), dim3(grid), dim3(512), 0, st, p);
#endif
      else if (halo_bna && g_bf3_halo_ilv) hipLaunchKernelGGL((conv3x3_bf3_halo_kernel<0, 1, true, 16, 13, 1>), dim3(grid), dim3(512), 0, st, p);
      else if (halo_bna) hipLaunchKernelGGL((conv3x3_bf3_halo_kernel<0, 1, true>), dim3(grid), dim3(512), 0, st, p);
      else if (p.a_raw && g_bf3_producers == 8 && g_bf3_slots == 6) hipLaunchKernelGGL((gemm_bf3_persist_ws_kernel<OPK_ROWK_BN, 0, 3, 1, 8, 6>), dim3(grid), dim3(768), 0, st, p);
      else if (p.a_raw && g_bf3_producers == 8 && !g_bf3_halo_ilv) hipLaunchKernelGGL((gemm_bf3_persist_ws_kernel<OPK_ROWK_BN, 0, 3, 1, 8, 4, 0>), dim3(grid), dim3(768), 0, st, p);
      else if (p.a_raw && g_bf3_producers == 8) hipLaunchKernelGGL((gemm_bf3_persist_ws_kernel<OPK_ROWK_BN, 0, 3, 1, 8>), dim3(grid), dim3(768), 0, st, p);
      else if (p.a_raw) hipLaunchKernelGGL((gemm_bf3_persist_ws_kernel<OPK_ROWK_BN, 0, 3, 1>), dim3(grid), dim3(512), 0, st, p);
      else if (halo) hipLaunchKernelGGL((conv3x3_bf3_halo_kernel<0, 1>), dim3(grid), dim3(512), 0, st, p);
      else if (im && !g_bf3_halo_ilv) hipLaunchKernelGGL((gemm_bf3_persist_ws_kernel<OPK_IM2COL, 0, 3, 1, 4, 4, 0>), dim3(grid), dim3(512), 0, st, p);
      else if (im) hipLaunchKernelGGL((gemm_bf3_persist_ws_kernel<OPK_IM2COL, 0, 3, 1>), dim3(grid), dim3(512), 0, st, p);
      else if (!g_bf3_halo_ilv) hipLaunchKernelGGL((gemm_bf3_persist_ws_kernel<OPK_ROWK, 0, 3, 1, 4, 4, 0>), dim3(grid), dim3(512), 0, st, p);
      else hipLaunchKernelGGL((gemm_bf3_persist_ws_kernel<OPK_ROWK, 0, 3, 1>), dim3(grid), dim3(512), 0, st, p);
    } else
    if (p.a_raw) hipLaunchKernelGGL((gemm_bf3_persist_ws_kernel<OPK_ROWK_BN>), dim3(grid), dim3(512), 0, st, p);
    else
    if (halo) hipLaunchKernelGGL(conv3x3_bf3_halo_kernel<0>, dim3(grid), dim3(512), 0, st, p);
    else if (im) hipLaunchKernelGGL((gemm_bf3_persist_ws_kernel<OPK_IM2COL>), dim3(grid), dim3(512), 0, st, p);
    else hipLaunchKernelGGL((gemm_bf3_persist_ws_kernel<OPK_ROWK>), dim3(grid), dim3(512), 0, st, p);
  }
  else if (persist && ws256 && !halo) {
    p.mtiles = ceil_div(p.M, 256); p.ntiles = ceil_div(p.N, 128);
    g_last_mtiles = ceil_div(p.M, 64);     // statistics rows per 64-row wave tile
    const int T4 = p.mtiles * p.ntiles, grid = ceil_div(T4, ceil_div(T4, g_bf3_persist_grid));
#ifdef DIC_EXPERIMENTS
    if (ws256_bna) hipLaunchKernelGGL((gemm_bf3_persist_ws256_kernel<OPK_ROWK, 1, true>), dim3(grid), dim3(768), 0, st, p);
    else
#endif
#ifdef DIC_EXPERIMENTS
    if (p.fmt == 1 && !im && g_bf3_ws256_ablate == 1) hipLaunchKernelGGL((gemm_bf3_persist_ws256_kernel<OPK_ROWK, 1, false, 1>), dim3(grid), dim3(768), 0, st, p);
    else if (p.fmt == 1 && !im && g_bf3_ws256_ablate == 2) hipLaunchKernelGGL((gemm_bf3_persist_ws256_kernel<OPK_ROWK, 1, false, 2>), dim3(grid), dim3(768), 0, st, p);
    else if (p.fmt == 1 && !im && g_bf3_ws256_ablate == 3) hipLaunchKernelGGL((gemm_bf3_persist_ws256_kernel<OPK_ROWK, 1, false, 3>), dim3(grid), dim3(768), 0, st, p);
    else if (p.fmt == 1 && !im && g_bf3_ws256_ablate == 4) hipLaunchKernelGGL((gemm_bf3_persist_ws256_kernel<OPK_ROWK, 1, false, 4>), dim3(grid), dim3(768), 0, st, p);
    else
#endif
    if (p.fmt == 1 && !im && g_bf3_halo_ilv) hipLaunchKernelGGL((gemm_bf3_persist_ws256_kernel<OPK_ROWK, 1, false, 0, 1>), dim3(grid), dim3(768), 0, st, p);
    else if (p.fmt == 1 && !im) hipLaunchKernelGGL((gemm_bf3_persist_ws256_kernel<OPK_ROWK, 1>), dim3(grid), dim3(768), 0, st, p);
#ifdef DIC_EXPERIMENTS
    else if (im) hipLaunchKernelGGL((gemm_bf3_persist_ws256_kernel<OPK_IM2COL>), dim3(grid), dim3(768), 0, st, p);
    else hipLaunchKernelGGL((gemm_bf3_persist_ws256_kernel<OPK_ROWK>), dim3(grid), dim3(768), 0, st, p);
#else
    else DIC_REQUIRE(false, "gemm_bf3: no 256x128 kernel for this operand kind / format in the product library");
#endif
  }
#ifdef DIC_EXPERIMENTS
  else if (persist) {                    // ablations of the product kernels, and the persistent kernel without producer waves
    g_last_mtiles = 2 * p.mtiles;
    const int grid = ceil_div(T, ceil_div(T, g_bf3_persist_grid));
    if (!halo && g_bf3_ws && g_bf3_ablate == 2 && !im) hipLaunchKernelGGL((gemm_bf3_persist_ws_kernel<OPK_ROWK, 0, 2>), dim3(grid), dim3(512), 0, st, p);
    else if (!halo && g_bf3_ws && g_bf3_ablate == 1 && !im) hipLaunchKernelGGL((gemm_bf3_persist_ws_kernel<OPK_ROWK, 1>), dim3(grid), dim3(512), 0, st, p);
    else if (!halo && g_bf3_ws && g_bf3_ablate == 4 && !im) hipLaunchKernelGGL((gemm_bf3_persist_ws_kernel<OPK_ROWK, 3>), dim3(grid), dim3(512), 0, st, p);
    else if (!halo && g_bf3_ws && g_bf3_ablate == 5 && !im) hipLaunchKernelGGL((gemm_bf3_persist_ws_kernel<OPK_ROWK, 4>), dim3(grid), dim3(512), 0, st, p);
    else if (!halo && g_bf3_ws && g_bf3_ablate == 6 && !im && p.fmt == 1) hipLaunchKernelGGL((gemm_bf3_persist_ws_kernel<OPK_ROWK, 6, 3, 1>), dim3(grid), dim3(512), 0, st, p);      // (56: f16x2, every other barrier skipped - timing only)
    else if (!halo && g_bf3_ws && g_bf3_ablate == 0 && !im && p.fmt == 1) hipLaunchKernelGGL((gemm_bf3_persist_ws_kernel<OPK_ROWK, 0, 3, 1>), dim3(grid), dim3(512), 0, st, p);
    else if (!halo && g_bf3_ws) { if (im) hipLaunchKernelGGL((gemm_bf3_persist_ws_kernel<OPK_IM2COL>), dim3(grid), dim3(512), 0, st, p);
                                  else hipLaunchKernelGGL((gemm_bf3_persist_ws_kernel<OPK_ROWK>), dim3(grid), dim3(512), 0, st, p); }
    else if (halo && g_bf3_ablate == 1) hipLaunchKernelGGL(conv3x3_bf3_halo_kernel<1>, dim3(grid), dim3(512), 0, st, p);
    else if (halo && g_bf3_ablate == 2) hipLaunchKernelGGL(conv3x3_bf3_halo_kernel<2>, dim3(grid), dim3(512), 0, st, p);
    else if (halo) hipLaunchKernelGGL(conv3x3_bf3_halo_kernel<3>, dim3(grid), dim3(512), 0, st, p);
    else if (im) hipLaunchKernelGGL((gemm_bf3_persist_kernel<OPK_IM2COL>), dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_bf3_persist_kernel<OPK_ROWK>), dim3(grid), dim3(256), 0, st, p);
  } else if (pipe && g_bf3_ablate > 0 && !im) {
    if (g_bf3_ablate == 1) hipLaunchKernelGGL((gemm_bf3_pipe_kernel<OPK_ROWK, 2, 1>), dim3(total), dim3(256), 0, st, p);
    else if (g_bf3_ablate == 2) hipLaunchKernelGGL((gemm_bf3_pipe_kernel<OPK_ROWK, 2, 2>), dim3(total), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_bf3_pipe_kernel<OPK_ROWK, 2, 3>), dim3(total), dim3(256), 0, st, p);
  } else if (pipe && g_bf3_stages == 3) {
    if (im) hipLaunchKernelGGL((gemm_bf3_pipe_kernel<OPK_IM2COL, 3>), dim3(total), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_bf3_pipe_kernel<OPK_ROWK, 3>), dim3(total), dim3(256), 0, st, p);
  } else if (pipe) {
    if (im) hipLaunchKernelGGL((gemm_bf3_pipe_kernel<OPK_IM2COL, 2>), dim3(total), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_bf3_pipe_kernel<OPK_ROWK, 2>), dim3(total), dim3(256), 0, st, p);
  } else if (tmv == 2 && tnv == 2) { if (im) launch_bf3_variant<OPK_IM2COL, 2, 2>(p, total, st); else launch_bf3_variant<OPK_ROWK, 2, 2>(p, total, st); }
  else if (!im && g_bf3_ablate == 1 && tmv == 1) hipLaunchKernelGGL((gemm_bf3_kernel<OPK_ROWK, 1, 1, 2, 1>), dim3(total), dim3(256), 0, st, p);
  else if (!im && g_bf3_ablate == 2 && tmv == 1) hipLaunchKernelGGL((gemm_bf3_kernel<OPK_ROWK, 1, 1, 2, 2>), dim3(total), dim3(256), 0, st, p);
#endif
  else if (tmv == 2) { if (im) launch_bf3_variant<OPK_IM2COL, 2, 1>(p, total, st); else launch_bf3_variant<OPK_ROWK, 2, 1>(p, total, st); }
  else { if (im) launch_bf3_variant<OPK_IM2COL, 1, 1>(p, total, st); else launch_bf3_variant<OPK_ROWK, 1, 1>(p, total, st); }
  DIC_LAUNCH_CHECK();
  gemm_profile_mark_end(st);
  if (rem128 > 0) {        // finish the remainder tiles of a persistent launch (quadrant-wise: the fix-up works on 64x64 tiles)
    GemmParams g{};
    g.M = p.M; g.N = p.N; g.K = p.K; g.ep = p.ep; g.mtiles = ceil_div(p.M, 64); g.ntiles = ceil_div(p.N, 64);      // (ep.alpha: 1, or the f16x2 unscale of the raw sums)
    g.tail_split = p.tail_split; g.tail_ws = p.tail_ws;
    g.tail128_first = p.tail_first_tile; g.tail128_ntiles = p.ntiles;
    DIC_TRY(gemm_launch_tail_fixup(g, rem128 * 4, st));
  }
  if (tail_tiles > 0) {
    GemmParams g{};
    g.M = p.M; g.N = p.N; g.K = p.K; g.ep = p.ep; g.mtiles = p.mtiles; g.ntiles = p.ntiles;
    g.tail_first_tile = p.tail_first_tile; g.tail_split = p.tail_split; g.tail_ws = p.tail_ws;
    if (bn_fuse && gemm_tail_fixup_bn_eligible(g, tail_tiles)) {      // fix-up + BatchNorm finalize in one launch
      DIC_TRY(gemm_launch_tail_fixup_bn(g, tail_tiles, *bn_fuse, st));
      if (bn_fused) *bn_fused = 1;
    } else {
      DIC_TRY(gemm_launch_tail_fixup(g, tail_tiles, st));
    }
  }
  return DIC_OK;
}

#ifdef DIC_EXPERIMENTS
// conv3-style 1x1 convolution on the A-stationary kernel (conv1x1_astat_bn_kernel): y_raw[M][CO] = relu(raw[M][C] * scale + shift) . W^T,
// f16x2 weights planes (scale in out_scale = 1 / (kF16ActScale * w_scale)); BatchNorm partials per 32-row wave tile: *mtiles_out rows
static int g_astat = 0;                  // codes 110 / 111 (experiments build): off (default) / on
void conv1x1_astat_switch(int on) { g_astat = on; }
bool conv1x1_astat_eligible(int M, int C, int CO) {
  return g_astat != 0 && (C == 128 || C == 256) && CO % 128 == 0 && CO >= 128 && M >= 64 && (long long)(CO + 1) * C * 2 < (1ll << 31);
}
int conv1x1_astat_bn(const float* raw, const float* scale, const float* shift, int relu, int M, int C, const unsigned short* const w_planes[3],
                     int CO, float* y, float* bn_partial, int* mtiles_out, hipStream_t st, float out_scale, unsigned* status) {
  DIC_REQUIRE(raw && scale && shift && y && w_planes && w_planes[0] && w_planes[1], "conv1x1_astat_bn: null pointer");
  if (!conv1x1_astat_eligible(M, C, CO)) return 1;
  Bf3Params p{};
  p.M = M; p.N = CO; p.K = C;
  p.B.p[0] = w_planes[0]; p.B.p[1] = w_planes[1]; p.B.p[2] = nullptr;
  p.B.kind = OPK_ROWK; p.B.ld = C; p.B.paired = 1;
  p.a_raw = raw; p.a_scale = scale; p.a_shift = shift; p.a_ld = C; p.a_relu = relu; p.status = status;
  p.ep = ep_store(y, CO, nullptr, ACT_NONE);
  p.ep.stats = bn_partial;
  p.ep.alpha = out_scale;
  p.fmt = 1;
  const int mt = ceil_div(M, 64);
  p.mtiles = mt; p.ntiles = CO / 128;
  gemm_profile_mark_begin(st, 2.0 * M * CO * (double)C, 3000 + OPK_ROWK_BN * 10 + 9,
                          4.0 * ((double)M * C + (double)CO * C + (double)M * CO));
  if (C == 256) hipLaunchKernelGGL(conv1x1_astat_bn_kernel<8>, dim3(mt), dim3(512), 0, st, p);
  else hipLaunchKernelGGL(conv1x1_astat_bn_kernel<4>, dim3(mt), dim3(512), 0, st, p);
  DIC_LAUNCH_CHECK();
  gemm_profile_mark_end(st);
  if (mtiles_out) *mtiles_out = 2 * mt;
  return DIC_OK;
}
#endif

// y_raw[B,OH,OW,CO] (fp32) = conv(x planes NHWC, w planes OHWI); BN partial sums like conv_fwd
int conv_fwd_bf3(const unsigned short* const x_planes[3], const ConvDesc& d, const unsigned short* const w_planes[3],
                 float* y, float* bn_partial, int* mtiles_out, float* tail_ws, hipStream_t st, const float* bias,
                 const BnFuseArgs* bn_fuse, int* bn_fused, int act, int tail_ws_slabs, int fmt, float out_scale,
                 const float* alpha_dev0, const float* alpha_dev1) {
  DIC_REQUIRE(!d.in_nchw && d.C % 32 == 0 && d.KH * d.KW <= 32, "conv_fwd_bf3: needs NHWC input with C %% 32 == 0");
  Bf3Params p{};
  p.M = d.M(); p.N = d.CO; p.K = d.K();
  for (int i = 0; i < 3; ++i) { p.A.p[i] = x_planes[i]; p.B.p[i] = w_planes[i]; }
  p.A.kind = (d.KH == 1 && d.KW == 1 && d.stride == 1 && d.pad == 0) ? OPK_ROWK : OPK_IM2COL;
  p.A.ld = d.C; p.A.g = d.geom(); p.A.paired = 1;
  p.B.kind = OPK_ROWK; p.B.ld = d.K(); p.B.paired = 1;
  p.ep = ep_store(y, d.CO, bias, act);
  p.ep.stats = bn_partial;
  p.fmt = fmt;
  if (fmt == 1) {      // 1 / (activation scale * weight scale): the f16x2 planes hold scaled values; factors chosen on the device come by pointer
    p.ep.alpha = out_scale; p.ep.alpha_dev[0] = alpha_dev0; p.ep.alpha_dev[1] = alpha_dev1;
  }
  if (bn_fused) *bn_fused = 0;
  DIC_TRY(launch_bf3(p, st, tail_ws, 1, nullptr, bn_fuse, bn_fused, tail_ws_slabs));
  if (mtiles_out) *mtiles_out = g_last_mtiles;
  return DIC_OK;
}

// 1x1 convolution whose input is formed on the fly: y_raw[M][CO] = act(raw[M][C] * scale[C] + shift[C] (+ res[M][C])) . W^T,
// i.e. the BatchNorm-apply (+ residual) + ReLU + three-plane split of the input happens in the producer waves of the
// persistent warp-specialised kernel instead of in a bn_apply_planes pass (20 B per element of HBM traffic and a launch less).
// act_out (nullable) receives the fp32 input values once.  Returns DIC_OK, 1 when the launch policy would not run this shape on
// that kernel (nothing launched: the caller takes the bn_apply_planes route), or a negative error.
// would conv1x1_fwd_bf3_bn run this shape (the launch policy's answer, nothing launched)?
bool conv1x1_bf3_bn_eligible(int M, int C, int CO, int tail_ws_slabs, int fmt) {
  if (C % 32 != 0 || C > kBnTabMax || (long long)M * C * 4 >= (1ll << 32)) return false;
  static float dummy;                         // stands for "a tail workspace is there"; never dereferenced
  Bf3Params p{};
  p.M = M; p.N = CO; p.K = C;
  p.A.kind = OPK_ROWK; p.A.ld = C; p.A.paired = 1;
  p.B.kind = OPK_ROWK; p.B.ld = C; p.B.paired = 1;
  p.a_raw = &dummy;
  p.fmt = fmt;
  p.ep = ep_store(&dummy, CO, nullptr, ACT_NONE);
  return launch_bf3(p, nullptr, &dummy, 1, nullptr, nullptr, nullptr, tail_ws_slabs, true) == DIC_OK;
}

int conv1x1_fwd_bf3_bn(const float* raw, const float* scale, const float* shift, const float* res, int relu, float* act_out,
                       int M, int C, const unsigned short* const w_planes[3], int CO, float* y, float* bn_partial,
                       int* mtiles_out, float* tail_ws, int tail_ws_slabs, hipStream_t st, const BnFuseArgs* bn_fuse, int* bn_fused, int fmt,
                       float out_scale, unsigned* status, const float* res_scale, const float* res_shift) {
  DIC_REQUIRE(!res_scale || (fmt == 1 && res && res_shift), "conv1x1_fwd_bf3_bn: a BatchNorm on the residual needs the f16x2 format, the residual and both tables");
  DIC_REQUIRE(raw && scale && shift && y && C % 32 == 0 && C <= kBnTabMax, "conv1x1_fwd_bf3_bn: C %% 32 == 0, C <= 2048");
  if ((long long)M * C * 4 >= (1ll << 32)) {                // the kernel addresses the input with 32-bit byte offsets
    set_last_error("conv1x1 with on-the-fly BatchNorm operand: input of %d x %d floats exceeds 4 GiB (32-bit offsets); nothing was launched", M, C);
    return 1;
  }
  Bf3Params p{};
  p.M = M; p.N = CO; p.K = C;
  for (int i = 0; i < 3; ++i) { p.A.p[i] = nullptr; p.B.p[i] = w_planes[i]; }
  p.A.kind = OPK_ROWK; p.A.ld = C; p.A.paired = 1;
  p.B.kind = OPK_ROWK; p.B.ld = C; p.B.paired = 1;
  p.a_raw = raw; p.a_scale = scale; p.a_shift = shift; p.a_res = res; p.a_out = act_out; p.a_ld = C; p.a_relu = relu;
  p.a_res_scale = res_scale; p.a_res_shift = res_shift;
  p.status = status;
  p.ep = ep_store(y, CO, nullptr, ACT_NONE);
  p.ep.stats = bn_partial;
  p.fmt = fmt;
  if (fmt == 1) p.ep.alpha = out_scale;      // 1 / (kF16ActScale * weight scale): the producer waves scale the activations by kF16ActScale
  if (bn_fused) *bn_fused = 0;
  const int rc = launch_bf3(p, st, tail_ws, 1, nullptr, bn_fuse, bn_fused, tail_ws_slabs);
  if (rc != DIC_OK) return rc;
  if (mtiles_out) *mtiles_out = g_last_mtiles;
  return DIC_OK;
}

// 3x3 / stride 1 / pad 1 convolution of 14x14 maps whose input is formed on the fly (f16x2 format): y_raw = conv(act(raw * scale[c] +
// shift[c])), the BatchNorm-apply + ReLU + split of the input done by the producer waves of the LDS-halo kernel (BNA form) instead of
// a bn_apply_planes pass.  Returns DIC_OK, 1 when the launch policy would not run this shape on that kernel (nothing launched: the
// caller takes the plane route), or a negative error.
int conv3x3_fwd_bf3_bn(const float* raw, const float* scale, const float* shift, int relu, const ConvDesc& d,
                       const unsigned short* const w_planes[3], float* y, float* bn_partial, int* mtiles_out, float* tail_ws,
                       int tail_ws_slabs, hipStream_t st, const BnFuseArgs* bn_fuse, int* bn_fused, int fmt, float out_scale, unsigned* status) {
  if (fmt != 1 || d.in_nchw || d.KH != 3 || d.KW != 3 || d.stride != 1 || d.pad != 1 || d.C % 32 != 0 || d.C > kHaloBnTab ||
      (long long)d.B * d.H * d.W * d.C * 4 >= (1ll << 32))
    return 1;
  DIC_REQUIRE(raw && scale && shift && y, "conv3x3_fwd_bf3_bn: null pointer");
  Bf3Params p{};
  p.M = d.M(); p.N = d.CO; p.K = d.K();
  for (int i = 0; i < 3; ++i) { p.A.p[i] = nullptr; p.B.p[i] = w_planes[i]; }
  p.A.kind = OPK_IM2COL; p.A.ld = d.C; p.A.g = d.geom(); p.A.paired = 1;
  p.B.kind = OPK_ROWK; p.B.ld = d.K(); p.B.paired = 1;
  p.a_raw = raw; p.a_scale = scale; p.a_shift = shift; p.a_ld = d.C; p.a_relu = relu;
  p.status = status;
  p.ep = ep_store(y, d.CO, nullptr, ACT_NONE);
  p.ep.stats = bn_partial;
  p.fmt = 1; p.ep.alpha = out_scale;
  if (bn_fused) *bn_fused = 0;
  const int rc = launch_bf3(p, st, tail_ws, 1, nullptr, bn_fuse, bn_fused, tail_ws_slabs);
  if (rc != DIC_OK) return rc;
  if (mtiles_out) *mtiles_out = g_last_mtiles;
  return DIC_OK;
}

// ------------------------------------------------------------------------------------------
// Weight gradient on the bf16x3 kernel: dW[co][(kh,kw,c)] = sum_m dY[m][co] * patch(m)[(kh,kw,c)] is a contraction over
// the output pixels m, so both operands are needed with m contiguous.  Two producers write them as paired planes
// (K = M rounded up to 32, zero filled): transpose_split (dY^T) and im2col_transpose_split (patch^T); the product then
// runs as a split-K launch (few output tiles, K in the ten thousands).
// One workgroup = a 32 (m) x 32 (column) tile through LDS; 16 consecutive threads write one 128-B plane line.
// ------------------------------------------------------------------------------------------
template <bool IM2COL>
__global__ void __launch_bounds__(256) transpose_split_kernel(const float* __restrict__ x, long long ld, int M, int Kpad,
                                                               int ncols, ConvGeom g, unsigned short* __restrict__ hi,
                                                               unsigned short* __restrict__ mid,
                                                               unsigned short* __restrict__ lo, float f16_scale,
                                                               const float* __restrict__ f16_slot) {      // lo == NULL: f16x2 planes of scale * x, scale = *f16_slot or f16_scale
  __shared__ float tile[32][33];
  const int m0 = blockIdx.x * 32;
  const int tid = threadIdx.x;
  int c0, tap = 0;                       // first source column of this tile (and the filter tap for im2col)
  if constexpr (IM2COL) {
    const int cb = g.C / 32;
    tap = blockIdx.y / cb;
    c0 = (blockIdx.y - tap * cb) * 32;
  } else {
    c0 = blockIdx.y * 32;
  }
  {  // load: thread (mrow, c4) reads 4 consecutive columns of one source row
    const int mrow = tid >> 3, c4 = (tid & 7) * 4;
    const int m = m0 + mrow;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < M) {
      if constexpr (IM2COL) {
        const int ohw = g.OH * g.OW;
        const int img = m / ohw, rem = m - img * ohw;
        const int oh = rem / g.OW, ow = rem - oh * g.OW;
        const int kh = tap / g.KW, kw = tap - kh * g.KW;
        const int ih = oh * g.stride - g.pad + kh, iw = ow * g.stride - g.pad + kw;
        if ((unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W)
          v = *reinterpret_cast<const float4*>(x + (((long long)img * g.H + ih) * g.W + iw) * g.C + c0 + c4);
      } else {
        if (c0 + c4 < ncols) v = *reinterpret_cast<const float4*>(x + (long long)m * ld + c0 + c4);
      }
    }
    tile[mrow][c4] = v.x; tile[mrow][c4 + 1] = v.y; tile[mrow][c4 + 2] = v.z; tile[mrow][c4 + 3] = v.w;
  }
  __syncthreads();
  {  // store: thread (q, parity, quad) writes 4 consecutive m of output row 2q + parity
    const int quad = tid & 7, parity = (tid >> 3) & 1, q = tid >> 4;
    const int c = 2 * q + parity;
    const long long row = (IM2COL ? (long long)tap * g.C : 0) + c0 + c;
    unsigned short h[4], mm[4], l[4];
    if (!lo) {
      const float fs = f16_slot ? f16_slot[0] : f16_scale;
#pragma unroll
      for (int j = 0; j < 4; ++j) split2_f16(tile[quad * 4 + j][c], fs, h[j], mm[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) split3_bf16(tile[quad * 4 + j][c], h[j], mm[j], l[j]);
    }
    const long long off = plane_offset(row, m0 + quad * 4, Kpad / 32, 1);
    *reinterpret_cast<uint2*>(hi + off) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
    *reinterpret_cast<uint2*>(mid + off) = make_uint2((unsigned)mm[0] | ((unsigned)mm[1] << 16), (unsigned)mm[2] | ((unsigned)mm[3] << 16));
    if (lo) *reinterpret_cast<uint2*>(lo + off) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
  }
}

size_t conv_wgrad_bf3_plane_elems(const ConvDesc& d, int which) {     // which: 0 = dY^T planes, 1 = patch^T planes
  const size_t kpad = ((size_t)d.M() + 31) / 32 * 32;
  const size_t rows = which == 0 ? (size_t)d.CO : (size_t)d.K();
  return ((rows + 1) & ~(size_t)1) * kpad;
}
// Few-tiles route of the weight gradient (round 4): dW is CO x K() with a very long contraction (every output pixel of the batch), e.g.
// 512 x 1152 over 30 976 for the depth encoder's conv2 - 36 tiles of 128 x 128.  launch_bf3's "few tiles, long K" form cuts every tile
// into K slices, one per workgroup of the persistent warp-specialised kernel, and the tail fix-up sums them in K order (deterministic):
// measured 190 -> ... us for conv2 against the 64 x 64 tiles with the caller's K split (720 workgroups of one computing wave per SIMD).
static bool wgrad_persist_shape(const ConvDesc& d) {
  const long long t22 = (long long)ceil_div(d.CO, 128) * ceil_div(d.K(), 128);
  return d.K() % 128 == 0 && t22 >= 32 && t22 < g_bf3_remainder_grid && (d.M() + 31) / 32 >= 64;
}
size_t conv_wgrad_bf3_ws_floats(const ConvDesc& d, int splitk) {
  size_t n = (size_t)ceil_div(d.CO, 64) * ceil_div(d.K(), 64) * splitk * 64 * 64;
  if (wgrad_persist_shape(d)) n = std::max(n, (size_t)4 * g_bf3_remainder_grid * 64 * 64);      // four [64][64] slabs per slice, <= one slice per CU
  return n;
}

int conv_wgrad_bf3(const float* x, const ConvDesc& d, const float* dy, float* dw_ohwi, int splitk,
                   unsigned short* const dyT[3], unsigned short* const pT[3], float* ws, hipStream_t st, int fmt, const float* dy_slot) {
  DIC_REQUIRE(!d.in_nchw && d.C % 32 == 0 && d.CO % 32 == 0, "conv_wgrad_bf3: NHWC input, C and CO %% 32");
  DIC_REQUIRE(fmt == 0 || dy_slot, "conv_wgrad_bf3: the f16x2 format needs the scale slot of the gradient");
  const int M = d.M(), Kpad = (M + 31) / 32 * 32;
  const ConvGeom g = d.geom();
  // f16x2: dY^T planes of (*dy_slot) * dY, patch^T planes of kF16ActScale * x (third plane pointer NULL tells the kernel the format)
  hipLaunchKernelGGL((transpose_split_kernel<false>), dim3(Kpad / 32, d.CO / 32), dim3(256), 0, st, dy, (long long)d.CO, M,
                     Kpad, d.CO, g, dyT[0], dyT[1], fmt ? nullptr : dyT[2], 1.0f, fmt ? dy_slot : nullptr);
  hipLaunchKernelGGL((transpose_split_kernel<true>), dim3(Kpad / 32, d.KH * d.KW * (d.C / 32)), dim3(256), 0, st, x,
                     (long long)d.C, M, Kpad, d.C, g, pT[0], pT[1], fmt ? nullptr : pT[2], kF16ActScale, (const float*)nullptr);
  DIC_LAUNCH_CHECK();
  Bf3Params p{};
  p.M = d.CO; p.N = d.K(); p.K = Kpad;
  for (int i = 0; i < 3; ++i) { p.A.p[i] = dyT[i]; p.B.p[i] = pT[i]; }
  if (fmt) p.A.p[2] = p.B.p[2] = nullptr;
  p.A.kind = OPK_ROWK; p.A.ld = Kpad; p.A.paired = 1;
  p.B.kind = OPK_ROWK; p.B.ld = Kpad; p.B.paired = 1;
  p.ep = ep_store(dw_ohwi, d.K(), nullptr, ACT_NONE);
  if (fmt) { p.fmt = 1; p.ep.alpha = 1.0f / kF16ActScale; p.ep.alpha_dev[0] = dy_slot + 1; }      // 1 / (s_dy * 4): the first factor lives on the device
  if (g_bf3_wgrad_persist && wgrad_persist_shape(d))
    return launch_bf3(p, st, ws, 1, nullptr, nullptr, nullptr, 4 * g_bf3_remainder_grid);
  return launch_bf3(p, st, nullptr, splitk, ws);
}

// ------------------------------------------------------------------------------------------
// 7x7 / stride-2 / pad-3 stem (C_in = 3) on the bf16x3 kernel.  The NCHW image is re-laid as zero-padded NHWC4 planes
// [B][H+6][Wp][4] (Wp = W + 8: 3 px left, 5 right), so that the 7 pixels x 3 channels of filter row kh seen by output
// (oh, ow) are the first 28 of 32 contiguous, 16-B aligned elements starting at padded pixel (2*oh + kh, 2*ow): one
// 64-byte strip = one K tile, K = 7 * 32 = 224 (the 8th pixel and the 4th channel meet zero weights).  Replaces the
// per-element gather of the register-staged kernel (48 TF-eq).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) stem_pack_image_kernel(const float* __restrict__ img, int B, int H, int W, int Hp,
                                                               int Wp, unsigned short* __restrict__ hi,
                                                               unsigned short* __restrict__ mid,
                                                               unsigned short* __restrict__ lo, unsigned* __restrict__ status) {      // lo == NULL: f16x2 planes of kF16ActScale * image (guarded)
  const long long total = (long long)B * Hp * Wp;         // one thread per padded pixel (4 channels = 8 B per plane)
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int pw = (int)(i % Wp);
    const long long r = i / Wp;
    const int ph = (int)(r % Hp);
    const long long b = r / Hp;
    const int h = ph - 3, w = pw - 3;
    unsigned short a[4] = {0, 0, 0, 0}, m[4] = {0, 0, 0, 0}, l[4] = {0, 0, 0, 0};
    if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float v = img[((b * 3 + c) * H + h) * W + w];
        if (lo) split3_bf16(v, a[c], m[c], l[c]);
        else {
          if (f16x2_out_of_range(v, kF16ActScale)) f16x2_raise(status, 16u);
          split2_f16(v, kF16ActScale, a[c], m[c]);
        }
      }
    }
    reinterpret_cast<uint2*>(hi)[i] = make_uint2((unsigned)a[0] | ((unsigned)a[1] << 16), (unsigned)a[2] | ((unsigned)a[3] << 16));
    reinterpret_cast<uint2*>(mid)[i] = make_uint2((unsigned)m[0] | ((unsigned)m[1] << 16), (unsigned)m[2] | ((unsigned)m[3] << 16));
    if (lo) reinterpret_cast<uint2*>(lo)[i] = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
  }
}

// stem weights OIHW [CO][3][7][7] -> fp32 [CO][7][8][4] (kw and channel zero-padded), the K order of the strips
__global__ void __launch_bounds__(256) stem_pack_weights_kernel(const float* __restrict__ w, int CO, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= CO * 224) return;
  const int c = i & 3, kw = (i >> 2) & 7, kh = (i >> 5) % 7, co = i / 224;
  out[i] = (c < 3 && kw < 7) ? w[((co * 3 + c) * 7 + kh) * 7 + kw] : 0.f;
}

size_t conv_stem_bf3_plane_elems(int B, int H, int W) { return (size_t)B * (H + 6) * (W + 8) * 4 + 64; }

int conv_stem_pack_weights(const float* w_oihw, int CO, float* scratch_f32, unsigned short* const w_planes[3], hipStream_t st, float f16_scale) {
  hipLaunchKernelGGL(stem_pack_weights_kernel, dim3(ceil_div(CO * 224, 256)), dim3(256), 0, st, w_oihw, CO, scratch_f32);
  DIC_LAUNCH_CHECK();
  if (f16_scale > 0.f) return split_f16x2_paired(scratch_f32, CO, 224, f16_scale, w_planes[0], w_planes[1], st, nullptr);      // (two planes of f16_scale * w)
  return split_bf16x3_paired(scratch_f32, CO, 224, w_planes[0], w_planes[1], w_planes[2], st);
}

// y_raw[B,OH,OW,CO] = conv7x7s2p3(imgs NCHW) with BN partial sums; x_planes: conv_stem_bf3_plane_elems(B,H,W) each
int conv_stem_bf3(const float* imgs_nchw, int B, int H, int W, int CO, unsigned short* const x_planes[3],
                  const unsigned short* const w_planes[3], float* y, float* bn_partial, int* mtiles_out, hipStream_t st, int fmt,
                  float out_scale, unsigned* status) {
  DIC_REQUIRE(H % 2 == 0 && W % 2 == 0, "conv_stem_bf3: even image sizes");
  const int Hp = H + 6, Wp = W + 8, OH = H / 2, OW = W / 2;
  const long long px = (long long)B * Hp * Wp;
  hipLaunchKernelGGL(stem_pack_image_kernel, dim3((unsigned)std::min<long long>((px + 255) / 256, 16384)), dim3(256), 0, st,
                     imgs_nchw, B, H, W, Hp, Wp, x_planes[0], x_planes[1], fmt ? nullptr : x_planes[2], status);
  DIC_LAUNCH_CHECK();
  Bf3Params p{};
  p.M = B * OH * OW; p.N = CO; p.K = 224;
  for (int i = 0; i < 3; ++i) { p.A.p[i] = x_planes[i]; p.B.p[i] = w_planes[i]; }
  if (fmt) { p.A.p[2] = p.B.p[2] = nullptr; p.fmt = 1; }
  p.A.kind = OPK_IM2COL; p.A.ld = 4; p.A.paired = 0;
  p.A.g = ConvGeom{Hp, Wp, 4, OH, OW, 7, 1, 2, 0, 2};        // nchw = 2: strip mode of the loader
  p.B.kind = OPK_ROWK; p.B.ld = 224; p.B.paired = 1;
  p.ep = ep_store(y, CO, nullptr, ACT_NONE);
  p.ep.stats = bn_partial;
  if (fmt) p.ep.alpha = out_scale;      // 1 / (kF16ActScale * weight scale)
  DIC_TRY(launch_bf3(p, st, nullptr));
  if (mtiles_out) *mtiles_out = g_last_mtiles;
  return DIC_OK;
}

int conv_dgrad_s1_bf3(const unsigned short* const dy_planes[3], const ConvDesc& d,
                      const unsigned short* const wflip_planes[3], float* dx, hipStream_t st, float* tail_ws, int tail_ws_slabs,
                      int fmt, const float* alpha_dev0, const float* alpha_dev1) {
  DIC_REQUIRE(d.stride == 1 && d.CO % 32 == 0, "conv_dgrad_s1_bf3: stride 1, CO %% 32");
  // full correlation of dY (an [OH,OW,CO] image) with the flipped kernel, padding KH-1-pad
  const ConvDesc dd{d.B, d.OH(), d.OW(), d.CO, d.C, d.KH, d.KW, 1, d.KH - 1 - d.pad, 0};
  return conv_fwd_bf3(dy_planes, dd, wflip_planes, dx, nullptr, nullptr, tail_ws, st, nullptr, nullptr, nullptr, ACT_NONE,
                      tail_ws_slabs, fmt, 1.0f, alpha_dev0, alpha_dev1);
}

int split_bf16x3_paired(const float* x, long long rows, int K, unsigned short* hi, unsigned short* mid,
                        unsigned short* lo, hipStream_t st) {
  DIC_REQUIRE(K % 32 == 0 && rows > 0, "split_bf16x3_paired: K %% 32");
  const long long n4 = ((rows + 1) >> 1) * (K / 2);
  const int blocks = (int)std::min<long long>((n4 + 255) / 256, 8192);
  hipLaunchKernelGGL(split_bf16x3_paired_kernel, dim3(blocks), dim3(256), 0, st, x, rows, K, hi, mid, lo);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

int split_f16x2_paired(const float* x, long long rows, int K, float scale, unsigned short* h1, unsigned short* h2, hipStream_t st,
                       unsigned* status) {
  DIC_REQUIRE(K % 32 == 0 && rows > 0 && scale > 0.f, "split_f16x2_paired: K %% 32, scale > 0");
  const long long n4 = ((rows + 1) >> 1) * (K / 2);
  const int blocks = (int)std::min<long long>((n4 + 255) / 256, 8192);
  hipLaunchKernelGGL(split_f16x2_paired_kernel, dim3(blocks), dim3(256), 0, st, x, rows, K, scale, h1, h2, status);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

int split_bf16x3(const float* x, long long n, unsigned short* hi, unsigned short* mid, unsigned short* lo, hipStream_t st) {
  const int blocks = (int)std::min<long long>((n + 255) / 256, 8192);
  hipLaunchKernelGGL(split_bf16x3_kernel, dim3(blocks), dim3(256), 0, st, x, n, hi, mid, lo);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

}  // namespace dic

using namespace dic;

extern "C" {

int dic_split_bf16x3(const float* x, long long n, uint16_t* hi, uint16_t* mid, uint16_t* lo, void* stream) {
  DIC_REQUIRE(x && hi && mid && lo && n > 0, "split_bf16x3: bad arguments");
  const int blocks = (int)std::min<long long>((n + 255) / 256, 8192);
  (void)blocks;
  return split_bf16x3(x, n, hi, mid, lo, (hipStream_t)stream);
}

int dic_split_bf16x3_paired(const float* x, long long rows, int K, uint16_t* hi, uint16_t* mid, uint16_t* lo,
                            void* stream) {
  DIC_REQUIRE(x && hi && mid && lo && rows > 0 && K > 0, "split_bf16x3_paired: bad arguments");
  return split_bf16x3_paired(x, rows, K, hi, mid, lo, (hipStream_t)stream);
}

int dic_split_f16x2_paired(const float* x, long long rows, int K, float scale, uint16_t* h1, uint16_t* h2, void* stream) {
  DIC_REQUIRE(x && h1 && h2 && rows > 0 && K > 0, "split_f16x2_paired: bad arguments");
  return split_f16x2_paired(x, rows, K, scale, h1, h2, (hipStream_t)stream);
}
int dic_split_f16x2_paired_checked(const float* x, long long rows, int K, float scale, uint16_t* h1, uint16_t* h2, uint32_t* overflow,
                                   void* stream) {
  DIC_REQUIRE(x && h1 && h2 && overflow && rows > 0 && K > 0, "split_f16x2_paired_checked: bad arguments");
  return split_f16x2_paired(x, rows, K, scale, h1, h2, (hipStream_t)stream, overflow);
}

static int gemm_bf16x3_any(int M, int N, int K, const uint16_t* a_hi, const uint16_t* a_mid, const uint16_t* a_lo,
                           long long lda, const uint16_t* b_hi, const uint16_t* b_mid, const uint16_t* b_lo, long long ldb,
                           float* C, long long ldc, const float* bias, int paired, void* stream) {
  DIC_REQUIRE(a_hi && a_mid && a_lo && b_hi && b_mid && b_lo && C, "gemm_bf16x3: null pointer");
  DIC_REQUIRE(M > 0 && N > 0 && K > 0 && K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "gemm_bf16x3: K, lda, ldb %% 8");
  Bf3Params p{};
  p.M = M; p.N = N; p.K = K;
  p.A.p[0] = a_hi; p.A.p[1] = a_mid; p.A.p[2] = a_lo; p.A.ld = lda; p.A.kind = OPK_ROWK;
  p.B.p[0] = b_hi; p.B.p[1] = b_mid; p.B.p[2] = b_lo; p.B.ld = ldb; p.B.kind = OPK_ROWK;
  p.A.paired = p.B.paired = paired;
  if (paired) DIC_REQUIRE(K % 32 == 0, "gemm_bf16x3_paired: K %% 32");
  p.ep = ep_store(C, ldc, bias, ACT_NONE);
  return launch_bf3(p, (hipStream_t)stream, nullptr);
}

/* y = act(x W^T + b) (+= with accumulate) and NHWC convolution with bias / activation on the split-bf16 kernels, operands as
 * paired planes (include/dic.h).  Used by the DPT front-end (dpt.py). */
int dic_linear_bf16x3(int M, int N, int K, const uint16_t* const x_planes[3], const uint16_t* const w_planes[3], const float* bias,
                      int act, int accumulate, float* C, long long ldc, void* stream) {
  DIC_REQUIRE(x_planes && w_planes && C && M > 0 && N > 0 && K > 0 && K % 32 == 0, "linear_bf16x3: bad arguments (K %% 32)");
  Bf3Params p{};
  p.M = M; p.N = N; p.K = K;
  for (int i = 0; i < 3; ++i) { p.A.p[i] = x_planes[i]; p.B.p[i] = w_planes[i]; }
  p.A.ld = K; p.A.kind = OPK_ROWK; p.B.ld = K; p.B.kind = OPK_ROWK;
  p.A.paired = p.B.paired = 1;
  p.ep = ep_store(C, ldc, bias, act);
  p.ep.accumulate = accumulate;
  return launch_bf3(p, (hipStream_t)stream, nullptr);
}
int dic_conv2d_bf16x3(const uint16_t* const x_planes[3], int B, int H, int W, int Cin, const uint16_t* const w_planes[3],
                      const float* bias, int CO, int KH, int KW, int stride, int pad, int act, float* y_nhwc, float* tail_ws,
                      void* stream) {
  DIC_REQUIRE(x_planes && w_planes && y_nhwc && Cin % 32 == 0, "conv2d_bf16x3: bad arguments (C %% 32)");
  ConvDesc d{B, H, W, Cin, CO, KH, KW, stride, pad, 0};
  return conv_fwd_bf3(x_planes, d, w_planes, y_nhwc, nullptr, nullptr, tail_ws, (hipStream_t)stream, bias, nullptr, nullptr, act);
}

/* the same two on the f16x2 operand format (planes from dic_split_f16x2_paired; out_scale = 1 / (scale of x * scale of W)) */
int dic_linear_f16x2(int M, int N, int K, const uint16_t* const x_planes[2], const uint16_t* const w_planes[2], const float* bias,
                     int act, int accumulate, float* C, long long ldc, float out_scale, void* stream) {
  DIC_REQUIRE(x_planes && w_planes && C && M > 0 && N > 0 && K > 0 && K % 32 == 0 && out_scale > 0.f, "linear_f16x2: bad arguments (K %% 32)");
  Bf3Params p{};
  p.M = M; p.N = N; p.K = K;
  for (int i = 0; i < 2; ++i) { p.A.p[i] = x_planes[i]; p.B.p[i] = w_planes[i]; }
  p.A.ld = K; p.A.kind = OPK_ROWK; p.B.ld = K; p.B.kind = OPK_ROWK;
  p.A.paired = p.B.paired = 1;
  p.ep = ep_store(C, ldc, bias, act);
  p.ep.accumulate = accumulate;
  p.fmt = 1; p.ep.alpha = out_scale;
  return launch_bf3(p, (hipStream_t)stream, nullptr);
}
int dic_conv2d_f16x2(const uint16_t* const x_planes[2], int B, int H, int W, int Cin, const uint16_t* const w_planes[2],
                     const float* bias, int CO, int KH, int KW, int stride, int pad, int act, float* y_nhwc, float* tail_ws,
                     float out_scale, void* stream) {
  DIC_REQUIRE(x_planes && w_planes && y_nhwc && Cin % 32 == 0 && out_scale > 0.f, "conv2d_f16x2: bad arguments (C %% 32)");
  ConvDesc d{B, H, W, Cin, CO, KH, KW, stride, pad, 0};
  const unsigned short* xp[3] = {x_planes[0], x_planes[1], nullptr};
  const unsigned short* wp[3] = {w_planes[0], w_planes[1], nullptr};
  return conv_fwd_bf3(xp, d, wp, y_nhwc, nullptr, nullptr, tail_ws, (hipStream_t)stream, bias, nullptr, nullptr, act, 256, 1, out_scale);
}

int dic_gemm_bf16x3(int M, int N, int K, const uint16_t* a_hi, const uint16_t* a_mid, const uint16_t* a_lo,
                    long long lda, const uint16_t* b_hi, const uint16_t* b_mid, const uint16_t* b_lo, long long ldb,
                    float* C, long long ldc, const float* bias, void* stream) {
  return gemm_bf16x3_any(M, N, K, a_hi, a_mid, a_lo, lda, b_hi, b_mid, b_lo, ldb, C, ldc, bias, 0, stream);
}

int dic_gemm_bf16x3_paired(int M, int N, int K, const uint16_t* a_hi, const uint16_t* a_mid, const uint16_t* a_lo,
                           const uint16_t* b_hi, const uint16_t* b_mid, const uint16_t* b_lo, float* C, long long ldc,
                           const float* bias, void* stream) {
  return gemm_bf16x3_any(M, N, K, a_hi, a_mid, a_lo, K, b_hi, b_mid, b_lo, K, C, ldc, bias, 1, stream);
}

}  // extern "C"
