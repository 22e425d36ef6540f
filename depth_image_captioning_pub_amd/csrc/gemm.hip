// Exact-fp32 MFMA GEMM / implicit-GEMM convolution kernel for gfx950 (see gemm.h).
#include "gemm.h"
#include "gemm_epilogue.h"
#include <algorithm>
#include <map>
#include <vector>

namespace dic {

constexpr int BK = 32;   // K tile (floats): one 128-B line per operand row

template <int KIND>
struct KFast { static constexpr bool v = (KIND == OPK_ROWK || KIND == OPK_IM2COL || KIND == OPK_GATHER); };

// ------------------------------------------------------------------------------------------
// Operand loaders: global -> registers (NV float4 per thread) -> LDS image S[k][i] (k-major).
//   K-fast kinds: thread -> (row = tid/8 + 32 r, k-chunk = tid%8); 8 lanes cover one 128-B line;
//                 LDS row stride BR+1 makes the transposing ds_write_b32 conflict-free.
//   I-fast kinds: thread -> (k = tid/(BR/4) + r*step, i4 = tid%(BR/4)); float4 along i,
//                 ds_write_b128 rows of stride BR+4.
// ------------------------------------------------------------------------------------------
template <int KIND, int BR>
struct Loader {
  static constexpr bool KFAST = KFast<KIND>::v;
  static constexpr int LD = KFAST ? BR + 1 : BR + 4;
  static constexpr int NV = BR / 32;
  static constexpr int I4 = BR / 4;          // threads per k row (I-fast)
  static constexpr int KSTEP = 256 / I4;     // k rows covered per pass (I-fast)

  const float* p;
  long long ld;
  int R, K, vec;
  ConvGeom g;
  int kc, i4, kk0, gi;
  const int* ktab;           // GATHER: LDS table k -> (input offset | kh << 20 | kw << 26), or null (see gather_table)
  bool valid[NV];
  long long base[NV];
  long long gbase[NV];       // GATHER table path: element offset of (image, ih0, iw0, channel 0)
  int ih0[NV], iw0[NV];
  int jkh[4], jkw[4], jc[4];
  bool jvalid[4];

  __device__ __forceinline__ void init(const GemmOperand& op, int r0, int R_, int K_) {
    p = op.p; ld = op.ld; R = R_; K = K_; vec = op.vec; g = op.g; ktab = nullptr;
    const int tid = threadIdx.x;
    if constexpr (KFAST) {
      kc = tid & 7;
#pragma unroll
      for (int r = 0; r < NV; ++r) {
        const int gr = r0 + (tid >> 3) + 32 * r;
        valid[r] = gr < R;
        if constexpr (KIND == OPK_ROWK) {
          base[r] = (long long)gr * ld + kc * 4;
        } else {
          const int ohw = g.OH * g.OW;
          const int img = gr / ohw, rem = gr - img * ohw;
          const int oh = rem / g.OW, ow = rem - oh * g.OW;
          ih0[r] = oh * g.stride - g.pad;
          iw0[r] = ow * g.stride - g.pad;
          base[r] = (KIND == OPK_IM2COL) ? (long long)img * g.H * g.W * g.C : (long long)img;
          gbase[r] = g.nchw ? (long long)img * g.C * g.H * g.W + (long long)ih0[r] * g.W + iw0[r]
                            : (((long long)img * g.H + ih0[r]) * g.W + iw0[r]) * g.C;
        }
      }
    } else {
      i4 = tid % I4;
      kk0 = tid / I4;
      gi = r0 + i4 * 4;
      if constexpr (KIND != OPK_COLK) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int gj = gi + j;
          jvalid[j] = gj < R;
          const int kpos = gj / g.C;
          jc[j] = gj - kpos * g.C;
          jkh[j] = kpos / g.KW;
          jkw[j] = kpos - jkh[j] * g.KW;
        }
      }
    }
  }

  __device__ __forceinline__ void load(int k0, float4 (&v)[NV]) const {
    if constexpr (KIND == OPK_ROWK) {
      const int k = k0 + kc * 4;
#pragma unroll
      for (int r = 0; r < NV; ++r) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid[r] && k < K) {
          const float* s = p + base[r] + k0;
          if (vec && k + 3 < K) {
            t = *reinterpret_cast<const float4*>(s);
          } else {
            t.x = s[0];
            if (k + 1 < K) t.y = s[1];
            if (k + 2 < K) t.z = s[2];
            if (k + 3 < K) t.w = s[3];
          }
        }
        v[r] = t;
      }
    } else if constexpr (KIND == OPK_IM2COL) {
      const int kpos = k0 / g.C, c0 = k0 - kpos * g.C;
      const int kh = kpos / g.KW, kw = kpos - kh * g.KW;
#pragma unroll
      for (int r = 0; r < NV; ++r) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        const int ih = ih0[r] + kh, iw = iw0[r] + kw;
        if (valid[r] && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W)
          t = *reinterpret_cast<const float4*>(p + base[r] + ((long long)ih * g.W + iw) * g.C + c0 + kc * 4);
        v[r] = t;
      }
    } else if constexpr (KIND == OPK_GATHER) {
#pragma unroll
      for (int r = 0; r < NV; ++r) {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = k0 + kc * 4 + j;
          float x = 0.f;
          if (ktab) {       // table path: no integer divisions per element (they made the 7x7 stems VALU-bound)
            if (valid[r] && k < K) {
              const int e = ktab[k];
              const int ih = ih0[r] + ((e >> 20) & 63), iw = iw0[r] + ((e >> 26) & 63);
              if ((unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W) x = p[gbase[r] + (e & 0xfffff)];
            }
          } else
          if (valid[r] && k < K) {
            const int kpos = k / g.C, c = k - kpos * g.C;
            const int kh = kpos / g.KW, kw = kpos - kh * g.KW;
            const int ih = ih0[r] + kh, iw = iw0[r] + kw;
            if ((unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W) {
              const long long img = base[r];
              const long long a = g.nchw ? ((img * g.C + c) * g.H + ih) * g.W + iw
                                         : ((img * g.H + ih) * g.W + iw) * g.C + c;
              x = p[a];
            }
          }
          e[j] = x;
        }
        v[r] = make_float4(e[0], e[1], e[2], e[3]);
      }
    } else if constexpr (KIND == OPK_COLK) {
#pragma unroll
      for (int r = 0; r < NV; ++r) {
        const int k = k0 + kk0 + r * KSTEP;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < K && gi < R) {
          const float* s = p + (long long)k * ld + gi;
          if (vec && gi + 3 < R) {
            t = *reinterpret_cast<const float4*>(s);
          } else {
            t.x = s[0];
            if (gi + 1 < R) t.y = s[1];
            if (gi + 2 < R) t.z = s[2];
            if (gi + 3 < R) t.w = s[3];
          }
        }
        v[r] = t;
      }
    } else {  // IM2COL_COLK / GATHER_COLK: i = (kh,kw,c) column, k = output pixel m
      const int ohw = g.OH * g.OW;
#pragma unroll
      for (int r = 0; r < NV; ++r) {
        const int m = k0 + kk0 + r * KSTEP;
        float e[4] = {0.f, 0.f, 0.f, 0.f};
        if (m < K) {
          const int img = m / ohw, rem = m - img * ohw;
          const int oh = rem / g.OW, ow = rem - oh * g.OW;
          if constexpr (KIND == OPK_IM2COL_COLK) {
            const int ih = oh * g.stride - g.pad + jkh[0], iw = ow * g.stride - g.pad + jkw[0];
            if (jvalid[0] && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W) {
              const float4 t = *reinterpret_cast<const float4*>(
                  p + (((long long)img * g.H + ih) * g.W + iw) * g.C + jc[0]);
              e[0] = t.x; e[1] = t.y; e[2] = t.z; e[3] = t.w;
            }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int ih = oh * g.stride - g.pad + jkh[j], iw = ow * g.stride - g.pad + jkw[j];
              if (jvalid[j] && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W) {
                const long long a = g.nchw ? (((long long)img * g.C + jc[j]) * g.H + ih) * g.W + iw
                                           : (((long long)img * g.H + ih) * g.W + iw) * g.C + jc[j];
                e[j] = p[a];
              }
            }
          }
        }
        v[r] = make_float4(e[0], e[1], e[2], e[3]);
      }
    }
  }

  __device__ __forceinline__ void store(float* S, const float4 (&v)[NV]) const {
    const int tid = threadIdx.x;
    if constexpr (KFAST) {
#pragma unroll
      for (int r = 0; r < NV; ++r) {
        const int row = (tid >> 3) + 32 * r;
        float* d = S + (kc * 4) * LD + row;
        d[0] = v[r].x;
        d[LD] = v[r].y;
        d[2 * LD] = v[r].z;
        d[3 * LD] = v[r].w;
      }
    } else {
#pragma unroll
      for (int r = 0; r < NV; ++r)
        *reinterpret_cast<float4*>(S + (kk0 + r * KSTEP) * LD + i4 * 4) = v[r];
    }
  }
};

// ------------------------------------------------------------------------------------------
// v2: both operands K-contiguous (linear layers, implicit-GEMM convolutions with C % 32 == 0).
// Tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no ds_write): one wave
// instruction fills 8 rows x 128 B of a row-major [rows][32] image.  The DMA destination is lane-linear,
// so the bank-conflict swizzle lives on the SOURCE side: 16-B position p of row r receives data chunk
// p ^ ((r>>1)&7); fragments are ds_read_b128 of chunk c at position c ^ ((r>>1)&7) (conflict-free for the
// 16-lane groups of ds_read_b128).  One 16-B read yields 4 k-steps; lane half h takes chunk 2g+h, i.e. the
// MFMA k index is permuted identically for A and B (k = 8g + 4h + s), which leaves the sum unchanged.
// Out-of-range rows / K tail / convolution padding read a zero line instead of being predicated.
// ------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(256))) const float g_zero_line[64] = {0.f};

template <int KIND, int BR>
struct DmaLoader {
  static constexpr int NI = BR / 32;          // wave-instructions per wave and K tile
  const float* p;
  int K, C, KW, W;
  long long lane_off[NI];                     // loop-invariant part of the source offset (floats)
  unsigned tapmask[NI];                       // IM2COL: bit (kh*KW+kw) set <=> that tap is inside the image
  bool valid[NI];
  int kchunk[NI];                             // ROWK: pre-swizzled chunk * 4 (for the K-tail test)

  // All per-lane address arithmetic happens here, once per output tile; per K tile only a wave-uniform
  // offset is added (the im2col tap moves every C/32 tiles).
  __device__ __forceinline__ void init(const GemmOperand& op, int r0, int R, int K_) {
    p = op.p; K = K_; C = op.g.C; KW = op.g.KW; W = op.g.W;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int n = 0; n < NI; ++n) {
      const int row = (n * 4 + w) * 8 + (lane >> 3);
      const int gr = r0 + row;
      valid[n] = gr < R;
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      kchunk[n] = chunk * 4;
      tapmask[n] = 0u;
      if constexpr (KIND == OPK_ROWK) {
        lane_off[n] = (long long)gr * op.ld + chunk * 4;
      } else {
        const ConvGeom& g = op.g;
        const int ohw = g.OH * g.OW;
        const int img = gr / ohw, rem = gr - img * ohw;
        const int oh = rem / g.OW, ow = rem - oh * g.OW;
        const int ih0 = oh * g.stride - g.pad, iw0 = ow * g.stride - g.pad;
        lane_off[n] = ((long long)img * g.H + ih0) * g.W * g.C + (long long)iw0 * g.C + chunk * 4;
        unsigned m = 0u;
        for (int kh = 0; kh < g.KH; ++kh)
          for (int kw = 0; kw < g.KW; ++kw)
            if ((unsigned)(ih0 + kh) < (unsigned)g.H && (unsigned)(iw0 + kw) < (unsigned)g.W) m |= 1u << (kh * g.KW + kw);
        tapmask[n] = valid[n] ? m : 0u;
      }
    }
  }

  // issue the DMA of K tile k0 into `img` (this operand's [BR][32] image of one stage)
  __device__ __forceinline__ void issue(int k0, float* img) const {
    const int w = threadIdx.x >> 6;
    int tap = 0;
    long long uni = k0;
    if constexpr (KIND == OPK_IM2COL) {
      tap = k0 / C;
      const int c0 = k0 - tap * C;
      const int kh = tap / KW, kw = tap - kh * KW;
      uni = ((long long)kh * W + kw) * C + c0;
    }
#pragma unroll
    for (int n = 0; n < NI; ++n) {
      bool ok;
      if constexpr (KIND == OPK_ROWK) ok = valid[n] && (k0 + kchunk[n] < K);
      else ok = (tapmask[n] >> tap) & 1u;
      const float* src = ok ? p + lane_off[n] + uni : g_zero_line;
      float* dst = img + ((n * 4 + w) * 8) * BK;            // wave-uniform 1-KiB block
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  }
};

template <int BM, int BN, int AK, int NSTAGE>
__global__ void __launch_bounds__(256) gemm_dma_kernel(const GemmParams p) {
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  constexpr int STAGE = (BM + BN) * BK;                        // floats per pipeline stage
  constexpr int NDMA = BM / 32 + BN / 32;                      // LDS-DMA instructions per wave and K tile
  __shared__ __align__(1024) float smem[NSTAGE * STAGE];       // ONE LDS object (guide: a second one de-pipelines)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles = p.mtiles * p.ntiles;
  const int nk = (p.K + BK - 1) / BK;
  int z, t, kt0, kt1, tail_slot = -1;
  if ((int)blockIdx.x < p.tail_first_block) {
    const int lb = xcd_remap(blockIdx.x, p.tail_first_block);
    z = lb / tiles;
    t = lb - z * tiles;
    kt0 = z * p.ktiles_per_split;
    kt1 = min(nk, kt0 + p.ktiles_per_split);
  } else {   // remainder tile, one K slice of it (spreads the last partial round over all CUs)
    const int q = (int)blockIdx.x - p.tail_first_block;
    const int piece = q % p.tail_split;
    t = p.tail_first_tile + q / p.tail_split;
    z = 0;
    const int per = (nk + p.tail_split - 1) / p.tail_split;
    kt0 = piece * per;
    kt1 = min(nk, kt0 + per);
    tail_slot = q;
  }
  const int tm = t / p.ntiles, tn = t - tm * p.ntiles;

  DmaLoader<AK, BM> la;
  DmaLoader<OPK_ROWK, BN> lbld;
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

  // prologue: NSTAGE-1 tiles in flight
#pragma unroll
  for (int s0 = 0; s0 < NSTAGE - 1; ++s0)
    if (s0 < nkt) {
      la.issue((kt0 + s0) * BK, smem + s0 * STAGE);
      lbld.issue((kt0 + s0) * BK, smem + s0 * STAGE + BM * BK);
    }

  // fragment addressing: row = wave offset + 32*t + (lane&31); swizzle key (row>>1)&7 depends on the lane only
  const int i31 = lane & 31, h = lane >> 5, key = (i31 >> 1) & 7;
  int pos[4];
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) pos[gq] = ((2 * gq + h) ^ key) * 4;
  const int aoff = (wm * WM + i31) * BK, boff = BM * BK + (wn * WN + i31) * BK;

  for (int it = 0; it < nkt; ++it) {
    // RAW: this wave's DMA of tile `it` has landed once at most the younger tiles' instructions are outstanding
    // (vmcnt counts in issue order); the barrier then covers the other waves' pieces.  WAR: the stage refilled
    // below was last read in iteration it-1, which every wave has left when it passes this barrier.
    const int younger = min(NSTAGE - 2, nkt - 1 - it);           // tiles issued after tile `it` and still wanted in flight
    if (NSTAGE == 2 || younger <= 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (younger == 1) {
      if constexpr (NDMA == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      if constexpr (NDMA == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    constexpr bool kAsmPath = (TM == 1 && TN == 1);
    const bool more = it + NSTAGE - 1 < nkt;
    float* nx = smem + ((it + NSTAGE - 1) % NSTAGE) * STAGE;
    if (!kAsmPath && more) {
      la.issue((kt0 + it + NSTAGE - 1) * BK, nx);
      lbld.issue((kt0 + it + NSTAGE - 1) * BK, nx + BM * BK);
    }
    if constexpr (kAsmPath) {
      // Fragment reads in inline asm: hipcc would otherwise drain the DMA ring (vmcnt(0)) before every ds_read
      // that follows an LDS-DMA issue, because it cannot prove the two do not alias.  The asm reads are not
      // tracked by the compiler, so lgkmcnt is counted here: issue order a0,b0,a1,b1,a2,b2,a3,b3; lgkmcnt(4)
      // retires the first four, the second statement (tied to the remaining registers) retires the rest.
      typedef float v4f __attribute__((ext_vector_type(4)));
      const unsigned sbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)smem +
                             (unsigned)((it % NSTAGE) * STAGE) * 4u;
      const unsigned aa = sbase + (unsigned)aoff * 4u, bb = sbase + (unsigned)boff * 4u;
      v4f a0, a1, a2, a3, b0, b1, b2, b3;
      asm volatile(
          "ds_read_b128 %0, %8\n\t"
          "ds_read_b128 %4, %12\n\t"
          "ds_read_b128 %1, %9\n\t"
          "ds_read_b128 %5, %13\n\t"
          "ds_read_b128 %2, %10\n\t"
          "ds_read_b128 %6, %14\n\t"
          "ds_read_b128 %3, %11\n\t"
          "ds_read_b128 %7, %15\n\t"
          "s_waitcnt lgkmcnt(4)"
          : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
          : "v"(aa + pos[0] * 4u), "v"(aa + pos[1] * 4u), "v"(aa + pos[2] * 4u), "v"(aa + pos[3] * 4u),
            "v"(bb + pos[0] * 4u), "v"(bb + pos[1] * 4u), "v"(bb + pos[2] * 4u), "v"(bb + pos[3] * 4u)
          : "memory");
      __builtin_amdgcn_sched_barrier(0);
#define DIC_MFMA4(A_, B_)                                                                     \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.x, B_.x, acc[0][0], 0, 0, 0);           \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.y, B_.y, acc[0][0], 0, 0, 0);           \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.z, B_.z, acc[0][0], 0, 0, 0);           \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A_.w, B_.w, acc[0][0], 0, 0, 0);
      DIC_MFMA4(a0, b0)
      if (more) la.issue((kt0 + it + NSTAGE - 1) * BK, nx);          // address math + DMA in the MFMA shadow
      DIC_MFMA4(a1, b1)
      if (more) lbld.issue((kt0 + it + NSTAGE - 1) * BK, nx + BM * BK);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a2), "+v"(a3), "+v"(b2), "+v"(b3)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      DIC_MFMA4(a2, b2)
      DIC_MFMA4(a3, b3)
#undef DIC_MFMA4
    } else {
    const float* st = smem + (it % NSTAGE) * STAGE;
    float4 af[TM][4], bf[TN][4];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i][gq] = *reinterpret_cast<const float4*>(st + aoff + i * 32 * BK + pos[gq]);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j][gq] = *reinterpret_cast<const float4*>(st + boff + j * 32 * BK + pos[gq]);
    }
#pragma unroll
    for (int gq = 0; gq < 4; ++gq)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][gq].x, bf[j][gq].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][gq].y, bf[j][gq].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][gq].z, bf[j][gq].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][gq].w, bf[j][gq].w, acc[i][j], 0, 0, 0);
        }
    }
  }
  __syncthreads();      // all fragment reads done before the epilogue reuses LDS as scratch
  if (tail_slot >= 0) {   // raw partial of this K slice, tile-local [BM][BN] layout
    float* dst = p.tail_ws + (long long)tail_slot * BM * BN;
    const int nl = wn * WN + (lane & 31), ml = wm * WM + 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          dst[(ml + i * 32 + (r & 3) + 8 * (r >> 2)) * BN + nl + j * 32] = acc[i][j][r];
    return;
  }
  gemm_epilogue<BM, BN>(p, acc, tm, tn, z, smem);
}

// Finishes the remainder tiles of a tail-split launch: sums the K-slice partials (fixed order), applies the epilogue
// and writes the BatchNorm column partials of that tile.  grid = remainder tiles; 1024 threads = 64 rows x 16 column
// quads.  The slices were written by other XCDs a moment ago, so every load is a long-latency miss: a thread issues
// all of its (<= 16) slice loads before the first add, which makes the kernel one memory round trip deep.
constexpr int kTailMaxSplit = 16;
template <int NS>      // NS = slice loads issued per thread (2/4/8/16 >= tail_split; the surplus re-reads the last slice)
__device__ __forceinline__ void tail_fixup_tile(const GemmParams& p, const int tail_index, float4 (*sred)[64][16],
                                                float4 (*sred2)[8][16]) {
  constexpr int BM = 64, BN = 64;
  int tm, tn;
  if (p.tail128_ntiles > 0) {     // a quadrant of a 128x128 remainder tile (persistent split-bf16 kernels)
    const int t128 = p.tail128_first + (tail_index >> 2), q = tail_index & 3;
    tm = 2 * (t128 / p.tail128_ntiles) + (q >> 1);
    tn = 2 * (t128 % p.tail128_ntiles) + (q & 1);
  } else {
    const int t = p.tail_first_tile + tail_index;
    tm = t / p.ntiles; tn = t - tm * p.ntiles;
  }
  const int c4 = threadIdx.x & 15, row = threadIdx.x >> 4;
  const float* base = p.tail_ws + (long long)tail_index * p.tail_split * BM * BN + row * BN + c4 * 4;
  float4 x[NS];
#pragma unroll
  for (int u = 0; u < NS; ++u) {     // branch-free guard: clamped slice, zeroed afterwards
    x[u] = *reinterpret_cast<const float4*>(base + (long long)min(u, p.tail_split - 1) * BM * BN);
    if (u >= p.tail_split) x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float4 v = x[0];
#pragma unroll
  for (int u = 1; u < NS; ++u) { v.x += x[u].x; v.y += x[u].y; v.z += x[u].z; v.w += x[u].w; }
  const int n = tn * BN + c4 * 4;
  const int m = tm * BM + row;
  float o[4] = {0.f, 0.f, 0.f, 0.f};
  if (m < p.M) {
    const float e[4] = {v.x, v.y, v.z, v.w};
    const float alpha = ep_alpha(p.ep);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (n + j < p.N) o[j] = finalize_store(p.ep, m, n + j, e[j], alpha);
  }
  if (p.ep.stats) {     // column sums / sums of squares over the tile's 64 rows: 8 groups of 8 rows, then the 8 groups
    sred[0][row][c4] = make_float4(o[0], o[1], o[2], o[3]);
    sred[1][row][c4] = make_float4(o[0] * o[0], o[1] * o[1], o[2] * o[2], o[3] * o[3]);
    __syncthreads();
    if (row < 16) {
      const int k = row >> 3, part = row & 7;
      float4 a = sred[k][part * 8][c4];
#pragma unroll
      for (int i = 1; i < 8; ++i) { const float4 b = sred[k][part * 8 + i][c4]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
      sred2[k][part][c4] = a;
    }
    __syncthreads();
    if (row < 2) {
      float4 a = sred2[row][0][c4];
#pragma unroll
      for (int i = 1; i < 8; ++i) { const float4 b = sred2[row][i][c4]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
      const float r[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n + j < p.N) p.ep.stats[((long long)tm * 2 + row) * p.N + n + j] = r[j];
    }
  }
}

template <int NS>
__global__ void __launch_bounds__(1024) tail_fixup_kernel(const GemmParams p) {
  __shared__ float4 sred[2][64][16];
  __shared__ float4 sred2[2][8][16];
  tail_fixup_tile<NS>(p, blockIdx.x, sred, sred2);
}

// Fix-up + BatchNorm finalize in one launch.  Workgroups [0, tail_tiles): the fix-up above.  Workgroups beyond: one
// per 32 channels; 1024 threads = 32 channels x 32 lanes.  Regular tile rows come from the partial table (fp64, lane-
// strided, all of a thread's loads in flight); the remainder tile rows are re-derived from the K slices (same slice
// order as the fix-up, so the same values), each lane taking rows lane and lane+32 of every remainder tile row.
template <int NS>
__global__ void __launch_bounds__(1024) tail_fixup_bn_kernel(const GemmParams p, const int tail_tiles, const BnFuseArgs bn) {
  __shared__ float4 sred[2][64][16];
  __shared__ float4 sred2[2][8][16];
  if ((int)blockIdx.x < tail_tiles) {
    tail_fixup_tile<NS>(p, blockIdx.x, sred, sred2);
    return;
  }
  double(*s1)[33] = reinterpret_cast<double(*)[33]>(&sred[0][0][0]);          // [32][33] doubles = 8448 B each
  double(*s2)[33] = reinterpret_cast<double(*)[33]>(&sred[1][0][0]);
  const int C = p.N;
  const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int c = ((int)blockIdx.x - tail_tiles) * 32 + cl;
  const int mt_reg = p.tail_first_tile / p.ntiles;                            // tile rows finished by the main kernel
  double a = 0.0, b = 0.0;
  if (c < C) {
    for (int t = g; t < mt_reg; t += 32 * 8) {
      float va[8], vb[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int tt = t + u * 32;
        const int tc = min(tt, mt_reg - 1);        // branch-free guard
        const float xa = p.ep.stats[((long long)tc * 2 + 0) * C + c], xb = p.ep.stats[((long long)tc * 2 + 1) * C + c];
        va[u] = tt < mt_reg ? xa : 0.f;
        vb[u] = tt < mt_reg ? xb : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { a += (double)va[u]; b += (double)vb[u]; }
    }
    // remainder tile rows: value = sum of the K slices (slice order), statistics straight in fp64
    const float tail_alpha = ep_alpha(p.ep);
    const int tn = c >> 6, col = c & 63;
    for (int tm = mt_reg; tm < p.mtiles; ++tm) {
      const long long q = (long long)(tm - mt_reg) * p.ntiles + tn;            // index among the remainder tiles
      const float* base = p.tail_ws + q * p.tail_split * 4096 + col;
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int row = g + 32 * rr;
        float x[NS];
#pragma unroll
        for (int u = 0; u < NS; ++u) {
          x[u] = base[(long long)min(u, p.tail_split - 1) * 4096 + row * 64];
          if (u >= p.tail_split) x[u] = 0.f;
        }
        float v = x[0];
#pragma unroll
        for (int u = 1; u < NS; ++u) v += x[u];
        v *= tail_alpha;         // as finalize_store (1, or the exact power-of-two unscale of the f16x2 operand format)
        if (tm * 64 + row < p.M) { a += (double)v; b += (double)v * (double)v; }
      }
    }
  }
  s1[g][cl] = a; s2[g][cl] = b;
  __syncthreads();
  if (g < 4) {
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
    const double mean = a / bn.count;
    double var = b / bn.count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = 1.0f / sqrtf((float)var + bn.eps);
    const float sc = bn.gamma[c] * invstd;
    bn.scale[c] = sc;
    bn.shift[c] = bn.beta[c] - (float)mean * sc;
    bn.mean[c] = (float)mean;
    bn.invstd[c] = invstd;
    const bool finite = fabs(a) <= 1.7e308 && fabs(b) <= 1.7e308;      // (as bn_finalize_train_kernel: the overflow guard's second line)
    if (!finite) f16x2_raise(bn.status, 8u);
    if (bn.running_mean && finite) {
      const double unb = bn.count > 1.0 ? var * bn.count / (bn.count - 1.0) : var;
      bn.running_mean[c] = fmaf(1.f - bn.momentum, bn.running_mean[c], __fmul_rn(bn.momentum, (float)mean));      // (form: bn_finalize_train_kernel, nn_kernels.hip)
      bn.running_var[c] = fmaf(1.f - bn.momentum, bn.running_var[c], __fmul_rn(bn.momentum, (float)unb));
    }
  }
}

// One output tile (or K slice of one) of a contraction: the body of gemm_kernel, also run by gemm_group_kernel over several problems.
// bid / nblk: this workgroup's index among the problem's nblk workgroups.
template <int BM, int BN, int AK, int BKIND, int ABL = 0>
__device__ __forceinline__ void gemm_tile_body(const GemmParams& p, const int bid, const int nblk) {
  using LA = Loader<AK, BM>;
  using LB = Loader<BKIND, BN>;
  constexpr int LDA = LA::LD, LDB = LB::LD;
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  __shared__ __align__(16) float As[2][BK * LDA];
  __shared__ __align__(16) float Bs[2][BK * LDB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lb = xcd_remap(bid, nblk);
  const int tiles = p.mtiles * p.ntiles;
  const int z = lb / tiles, t = lb - z * tiles;
  const int tm = t / p.ntiles, tn = t - tm * p.ntiles;

  LA la;
  LB lbld;
  la.init(p.A, tm * BM, p.M, p.K);
  lbld.init(p.B, tn * BN, p.N, p.K);
  if constexpr (AK == OPK_GATHER) {
    // k -> (kh, kw, c) decomposition once per workgroup instead of two integer divisions per gathered element
    __shared__ int ktab[256];
    const ConvGeom& g = p.A.g;
    if (p.K <= 256 && (long long)g.C * g.H * g.W < (1 << 20) && g.KH < 64 && g.KW < 64) {
      if (tid < p.K) {
        const int kpos = tid / g.C, c = tid - kpos * g.C;
        const int kh = kpos / g.KW, kw = kpos - kh * g.KW;
        const int off = g.nchw ? (c * g.H + kh) * g.W + kw : (kh * g.W + kw) * g.C + c;
        ktab[tid] = off | (kh << 20) | (kw << 26);
      }
      __syncthreads();
      la.ktab = ktab;
    }
  }

  const int nk = (p.K + BK - 1) / BK;
  const int kt0 = z * p.ktiles_per_split;
  const int kt1 = min(nk, kt0 + p.ktiles_per_split);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Register-staged pipeline, two tiles deep: at iteration `it` the registers hold tile it+1 (loaded one
  // iteration ago, so its latency is covered by a whole tile of MFMAs); it is written to the free LDS
  // buffer at the top of the iteration, the loads for tile it+2 are issued, and only then the MFMAs of
  // tile `it` run.  All fragments of a K tile are read from LDS up front (one wait ladder instead of a
  // read->wait->MFMA round trip per k-step).
  float4 ra[LA::NV], rb[LB::NV];
  const int nkt = kt1 - kt0;
  if (nkt > 0) {
    la.load(kt0 * BK, ra);
    lbld.load(kt0 * BK, rb);
    la.store(As[0], ra);
    lbld.store(Bs[0], rb);
  }
  if (nkt > 1) {
    la.load((kt0 + 1) * BK, ra);
    lbld.load((kt0 + 1) * BK, rb);
  }
  __syncthreads();

  const int arow = wm * WM + (lane & 31);
  const int brow = wn * WN + (lane & 31);
  const int khalf = lane >> 5;
  for (int it = 0; it < nkt; ++it) {
    const int cur = it & 1;
    if (ABL == 0 || ABL == 3) {
    if (it + 1 < nkt) {
      la.store(As[cur ^ 1], ra);
      lbld.store(Bs[cur ^ 1], rb);
    }
    if (it + 2 < nkt && ABL == 0) {
      la.load((kt0 + it + 2) * BK, ra);
      lbld.load((kt0 + it + 2) * BK, rb);
    }
    }
    const float* Ac = As[cur] + khalf * LDA + arow;
    const float* Bc = Bs[cur] + khalf * LDB + brow;
    float af[TM][BK / 2], bf[TN][BK / 2];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i][kk] = (ABL == 2) ? (float)(it + kk + i) : Ac[kk * 2 * LDA + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j][kk] = (ABL == 2) ? (float)(it - kk + j) : Bc[kk * 2 * LDB + j * 32];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][kk], bf[j][kk], acc[i][j], 0, 0, 0);
    __syncthreads();
  }

  gemm_epilogue<BM, BN>(p, acc, tm, tn, z, As[0]);
}

template <int BM, int BN, int AK, int BKIND, int ABL = 0>
__global__ void __launch_bounds__(256) gemm_kernel(const GemmParams p) {
  gemm_tile_body<BM, BN, AK, BKIND, ABL>(p, blockIdx.x, gridDim.x);
}

// Several INDEPENDENT contractions of one operand-kind pair in one launch: workgroup b belongs to problem i with
// first[i] <= b < first[i + 1].  For the batches of small products that follow a dependent chain (the decoder's weight gradients after
// BPTT: five launches of 5-70 us, each with its own launch boundary, partly filled grid and tail) - one grid, one tail.
struct GemmGroup {
  int n;
  int first[kGemmGroupMax + 1];
  GemmParams p[kGemmGroupMax];
};
static_assert(sizeof(GemmGroup) <= 4096, "GemmGroup travels as a kernel argument");
template <int BM, int BN, int AK, int BKIND>
__global__ void __launch_bounds__(256) gemm_group_kernel(const GemmGroup g) {
  int i = 0;
#pragma unroll
  for (int j = 1; j < kGemmGroupMax; ++j)
    if (j < g.n && (int)blockIdx.x >= g.first[j]) i = j;
  gemm_tile_body<BM, BN, AK, BKIND>(g.p[i], (int)blockIdx.x - g.first[i], g.first[i + 1] - g.first[i]);
}

__device__ __forceinline__ void splitk_reduce_body(const GemmParams& p, const int bid, const int nblk) {
  const long long total = (long long)p.M * p.N;
  const float alpha = ep_alpha(p.ep);
  for (long long e = (long long)bid * 256 + threadIdx.x; e < total; e += (long long)nblk * 256) {
    float s = 0.f;
    for (int z = 0; z < p.splitk; ++z) s += p.ws[(long long)z * total + e];
    const int m = (int)(e / p.N), n = (int)(e - (long long)m * p.N);
    finalize_store(p.ep, m, n, s, alpha);
  }
}
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const GemmParams p) { splitk_reduce_body(p, blockIdx.x, gridDim.x); }
__global__ void __launch_bounds__(256) splitk_reduce_group_kernel(const GemmGroup g) {      // first[] counts reduce workgroups here
  int i = 0;
#pragma unroll
  for (int j = 1; j < kGemmGroupMax; ++j)
    if (j < g.n && (int)blockIdx.x >= g.first[j]) i = j;
  splitk_reduce_body(g.p[i], (int)blockIdx.x - g.first[i], g.first[i + 1] - g.first[i]);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// Optional per-launch timing (bench.py roofline): HIP events recorded on the launch stream around
// every contraction launch, summed per kernel instantiation (tile, A kind, B kind).
struct ProfRec { hipEvent_t e0, e1; double flops; int key; double bytes; };
static bool g_prof_on = false;
static bool g_force_v1 = false;   // benchmarking switch: register-staged kernel for every shape
static int g_dma_stages = 2;
static bool g_tail_split_on = true;
static bool g_tail_skip_fix = false;   // timing experiment only
void gemm_force_v1(int on) {
  g_force_v1 = (on == 1);
  if (on == 2 || on == 3) g_dma_stages = on;
  if (on == 10) g_tail_split_on = false;
  if (on == 11) g_tail_split_on = true;
  if (on == 12) g_tail_skip_fix = true;
  if (on == 13) g_tail_skip_fix = false;
}
static std::vector<ProfRec> g_prof_recs;
static std::vector<hipEvent_t> g_prof_pool;

static hipEvent_t prof_event() {
  if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

// hooks for contraction launches that do not go through gemm_launch (gemm_bf3.hip)
static ProfRec g_open_rec{};
void gemm_profile_mark_begin(hipStream_t st, double flops, int key, double bytes) {
  if (!g_prof_on) return;
  g_open_rec.e0 = prof_event(); g_open_rec.e1 = prof_event();
  g_open_rec.flops = flops; g_open_rec.key = key; g_open_rec.bytes = bytes;
  (void)hipEventRecord(g_open_rec.e0, st);
}
void gemm_profile_mark_end(hipStream_t st) {
  if (!g_prof_on) return;
  (void)hipEventRecord(g_open_rec.e1, st);
  g_prof_recs.push_back(g_open_rec);
}

int gemm_profile_begin() {
  for (auto& r : g_prof_recs) { g_prof_pool.push_back(r.e0); g_prof_pool.push_back(r.e1); }
  g_prof_recs.clear();
  g_prof_on = true;
  return DIC_OK;
}

int gemm_profile_end(int max_entries, int* keys, double* total_ms, double* total_flops, long long* launches, int* n_out, double* total_bytes) {
  g_prof_on = false;
  DIC_CHECK_HIP(hipDeviceSynchronize());
  std::map<int, int> slot;
  int n = 0;
  for (auto& r : g_prof_recs) {
    float ms = 0.f;
    DIC_CHECK_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
    auto it = slot.find(r.key);
    int i;
    if (it == slot.end()) {
      if (n >= max_entries) continue;
      i = n++; slot[r.key] = i; keys[i] = r.key; total_ms[i] = 0; total_flops[i] = 0; launches[i] = 0;
      if (total_bytes) total_bytes[i] = 0;
    } else i = it->second;
    total_ms[i] += ms; total_flops[i] += r.flops; launches[i] += 1;
    if (total_bytes) total_bytes[i] += r.bytes;
  }
  for (auto& r : g_prof_recs) { g_prof_pool.push_back(r.e0); g_prof_pool.push_back(r.e1); }
  g_prof_recs.clear();
  *n_out = n;
  return DIC_OK;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

GemmOperand op_rowk(const float* p, long long ld) {
  GemmOperand o{};
  o.p = p; o.ld = ld; o.kind = OPK_ROWK; o.vec = aligned16(p) && (ld % 4 == 0);
  return o;
}
GemmOperand op_colk(const float* p, long long ld) {
  GemmOperand o{};
  o.p = p; o.ld = ld; o.kind = OPK_COLK; o.vec = aligned16(p) && (ld % 4 == 0);
  return o;
}
GemmOperand op_im2col(const float* x, const ConvGeom& g) {
  GemmOperand o{};
  o.p = x; o.ld = 0; o.kind = OPK_IM2COL; o.vec = 1; o.g = g;
  return o;
}
GemmOperand op_gather(const float* x, const ConvGeom& g) {
  GemmOperand o{};
  o.p = x; o.ld = 0; o.kind = OPK_GATHER; o.vec = 0; o.g = g;
  return o;
}
GemmOperand op_im2col_colk(const float* x, const ConvGeom& g) {
  GemmOperand o{};
  o.p = x; o.ld = 0; o.kind = OPK_IM2COL_COLK; o.vec = 1; o.g = g;
  return o;
}
GemmOperand op_gather_colk(const float* x, const ConvGeom& g) {
  GemmOperand o{};
  o.p = x; o.ld = 0; o.kind = OPK_GATHER_COLK; o.vec = 0; o.g = g;
  return o;
}
GemmEpilogue ep_store(float* C, long long ldc, const float* bias, int act) {
  GemmEpilogue e{};
  e.C = C; e.ldc = ldc; e.bias = bias; e.act = act; e.alpha = 1.0f;
  return e;
}

size_t gemm_splitk_ws_bytes(int M, int N, int splitk) {
  return splitk > 1 ? (size_t)splitk * M * N * sizeof(float) : 0;
}

int gemm_pick_tile(int M, int N) {
  // Measured on MI355X (scripts/bench_gemm.py): the 64x64 tile (4 workgroups/CU) matches or beats 128x128 on
  // every shape of the path up to 4096^3; the big tile only pays once the grid is many rounds deep.
  const long long t128 = (long long)ceil_div(M, 128) * ceil_div(N, 128);
  if (N >= 128 && M >= 128 && t128 >= 4096) return 128;
  return 64;
}

template <int BM, int BN>
static int launch_tile(const GemmParams& p, hipStream_t st) {
  const dim3 grid(p.mtiles * p.ntiles * p.splitk), block(256);
  const int a = p.A.kind, b = p.B.kind;
#define DIC_GEMM_CASE(AK_, BK_)                                                     \
  if (a == AK_ && b == BK_) {                                                       \
    hipLaunchKernelGGL((gemm_kernel<BM, BN, AK_, BK_>), grid, block, 0, st, p);     \
    return DIC_OK;                                                                  \
  }
  if (a == OPK_ROWK && b == OPK_ROWK && p.ablate > 0) {
    if (p.ablate == 1) hipLaunchKernelGGL((gemm_kernel<BM, BN, OPK_ROWK, OPK_ROWK, 1>), grid, block, 0, st, p);
    else if (p.ablate == 2) hipLaunchKernelGGL((gemm_kernel<BM, BN, OPK_ROWK, OPK_ROWK, 2>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gemm_kernel<BM, BN, OPK_ROWK, OPK_ROWK, 3>), grid, block, 0, st, p);
    return DIC_OK;
  }
  DIC_GEMM_CASE(OPK_ROWK, OPK_ROWK)
  DIC_GEMM_CASE(OPK_ROWK, OPK_COLK)
  DIC_GEMM_CASE(OPK_COLK, OPK_COLK)
  DIC_GEMM_CASE(OPK_COLK, OPK_ROWK)
  DIC_GEMM_CASE(OPK_IM2COL, OPK_ROWK)
  DIC_GEMM_CASE(OPK_GATHER, OPK_ROWK)
  DIC_GEMM_CASE(OPK_COLK, OPK_IM2COL_COLK)
  DIC_GEMM_CASE(OPK_COLK, OPK_GATHER_COLK)
#undef DIC_GEMM_CASE
  set_last_error("gemm: unsupported operand kinds A=%d B=%d", a, b);
  return DIC_ERR_UNSUPPORTED;
}

int gemm_launch_tail_fixup(const GemmParams& p, int tail_tiles, hipStream_t st) {
  const dim3 g(tail_tiles), b(1024);
  if (p.tail_split <= 2) hipLaunchKernelGGL(tail_fixup_kernel<2>, g, b, 0, st, p);
  else if (p.tail_split <= 4) hipLaunchKernelGGL(tail_fixup_kernel<4>, g, b, 0, st, p);
  else if (p.tail_split <= 8) hipLaunchKernelGGL(tail_fixup_kernel<8>, g, b, 0, st, p);
  else hipLaunchKernelGGL(tail_fixup_kernel<16>, g, b, 0, st, p);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

bool gemm_tail_fixup_bn_eligible(const GemmParams& p, int tail_tiles) {
  return tail_tiles > 0 && p.ep.stats != nullptr && p.tail_first_tile % p.ntiles == 0 && p.ep.bias == nullptr &&
         p.ep.act == ACT_NONE && !p.ep.accumulate && !p.ep.row_map && !p.ep.C2 && p.N % 32 == 0 &&
         p.mtiles <= 512;     // (larger layers use the two-stage finalize)
}

int gemm_launch_tail_fixup_bn(const GemmParams& p, int tail_tiles, const BnFuseArgs& bn, hipStream_t st) {
  const dim3 g(tail_tiles + p.N / 32), b(1024);
  if (p.tail_split <= 2) hipLaunchKernelGGL(tail_fixup_bn_kernel<2>, g, b, 0, st, p, tail_tiles, bn);
  else if (p.tail_split <= 4) hipLaunchKernelGGL(tail_fixup_bn_kernel<4>, g, b, 0, st, p, tail_tiles, bn);
  else if (p.tail_split <= 8) hipLaunchKernelGGL(tail_fixup_bn_kernel<8>, g, b, 0, st, p, tail_tiles, bn);
  else hipLaunchKernelGGL(tail_fixup_bn_kernel<16>, g, b, 0, st, p, tail_tiles, bn);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

int gemm_launch(GemmParams p, hipStream_t st, int force_tile) {
  DIC_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm: bad shape %d %d %d", p.M, p.N, p.K);
  DIC_REQUIRE(p.A.p && p.B.p && p.ep.C, "gemm: null pointer");
  if (p.A.kind == OPK_IM2COL) DIC_REQUIRE(p.A.g.C % 32 == 0, "im2col loader needs C %% 32 == 0 (C=%d)", p.A.g.C);
  const bool taps_ok = p.A.kind != OPK_IM2COL || p.A.g.KH * p.A.g.KW <= 32;      // v2 keeps a per-tap bitmask
  if (p.B.kind == OPK_IM2COL_COLK) DIC_REQUIRE(p.B.g.C % 4 == 0, "im2col_colk loader needs C %% 4 == 0");
  const int tile = force_tile ? force_tile : gemm_pick_tile(p.M, p.N);
  p.mtiles = ceil_div(p.M, tile);
  p.ntiles = ceil_div(p.N, tile);
  const int nk = ceil_div(p.K, BK);
  if (p.splitk < 1) p.splitk = 1;
  if (p.splitk > nk) p.splitk = nk;
  p.ktiles_per_split = ceil_div(nk, p.splitk);
  p.splitk = ceil_div(nk, p.ktiles_per_split);     // no empty slices
  if (p.splitk > 1) {
    DIC_REQUIRE(p.ws != nullptr, "gemm: split-K needs a workspace");
    DIC_REQUIRE(p.ep.stats == nullptr, "gemm: BN statistics epilogue cannot be combined with split-K");
  }
  if (p.ep.alpha == 0.0f) p.ep.alpha = 1.0f;
  const bool dma_ok = (p.B.kind == OPK_ROWK && p.B.vec && (p.K % 4 == 0) && p.ablate == 0 && !g_force_v1 && taps_ok &&
                       (p.A.kind == OPK_IM2COL || (p.A.kind == OPK_ROWK && p.A.vec)));
  // Tail plan (v2, 64-tile, no split-K): T tiles on 256 CUs leave r = T mod 256 tiles for a last partial round;
  // when r is small those tiles are cut into s = floor(256/r) K slices (<= 256 extra workgroups of 1/s tile each)
  // so the remainder spreads over the whole chip instead of costing a full round (measured: 784 tiles cost as much
  // as 1024 without this).
  int total_blocks = p.mtiles * p.ntiles * p.splitk;
  p.tail_first_block = total_blocks; p.tail_first_tile = 0; p.tail_split = 1;
  int tail_tiles = 0;
  if (dma_ok && tile == 64 && p.splitk == 1 && p.tail_ws != nullptr && g_tail_split_on) {
    const int T = p.mtiles * p.ntiles, r = T % 256;
    int s = r > 0 ? 256 / r : 0;
    s = std::min(s, std::min(nk / 2, kTailMaxSplit));
    // worth a fix-up launch (~8 us) only on shallow grids, where one partial round is a big share of the time
    if (r > 0 && r <= 128 && s >= 2 && T < 7 * 256) {
      tail_tiles = r;
      p.tail_first_tile = T - r;
      p.tail_first_block = T - r;
      p.tail_split = s;
      total_blocks = (T - r) + r * s;
    }
  }
  if (tail_tiles == 0) p.tail_ws = nullptr;
  ProfRec rec{};
  if (g_prof_on) {
    rec.e0 = prof_event(); rec.e1 = prof_event();
    rec.flops = 2.0 * p.M * p.N * (double)p.K;
    rec.bytes = 4.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N);      // operands read once + output written once (fp32)
    rec.key = (dma_ok ? 1000 : 0) + (tile == 128 ? 100 : 0) + p.A.kind * 10 + p.B.kind;
    (void)hipEventRecord(rec.e0, st);
  }
  int rc = DIC_OK;
  if (dma_ok) {
    const dim3 grid(total_blocks), block(256);
    if (tile == 128) {
      if (p.A.kind == OPK_IM2COL) hipLaunchKernelGGL((gemm_dma_kernel<128, 128, OPK_IM2COL, 2>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((gemm_dma_kernel<128, 128, OPK_ROWK, 2>), grid, block, 0, st, p);
    } else if (p.raw_partials && p.A.kind == OPK_ROWK && p.mtiles == 1 && p.ktiles_per_split <= 8) {
      // skinny split-K slabs (the decoder's per-step GEMMs, M <= 64): a 4-deep ring puts (nearly) the whole K slice
      // of a workgroup in flight at once - these launches are latency-, not throughput-bound
      hipLaunchKernelGGL((gemm_dma_kernel<64, 64, OPK_ROWK, 4>), grid, block, 0, st, p);
    } else if (g_dma_stages == 2) {
      if (p.A.kind == OPK_IM2COL) hipLaunchKernelGGL((gemm_dma_kernel<64, 64, OPK_IM2COL, 2>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((gemm_dma_kernel<64, 64, OPK_ROWK, 2>), grid, block, 0, st, p);
    } else {
      if (p.A.kind == OPK_IM2COL) hipLaunchKernelGGL((gemm_dma_kernel<64, 64, OPK_IM2COL, 3>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((gemm_dma_kernel<64, 64, OPK_ROWK, 3>), grid, block, 0, st, p);
    }
  } else {
    rc = (tile == 128) ? launch_tile<128, 128>(p, st) : launch_tile<64, 64>(p, st);
  }
  if (rc != 0) return rc;
  DIC_LAUNCH_CHECK();
  if (tail_tiles > 0 && !g_tail_skip_fix) {
    const dim3 g(tail_tiles), b(1024);
  if (p.tail_split <= 2) hipLaunchKernelGGL(tail_fixup_kernel<2>, g, b, 0, st, p);
  else if (p.tail_split <= 4) hipLaunchKernelGGL(tail_fixup_kernel<4>, g, b, 0, st, p);
  else if (p.tail_split <= 8) hipLaunchKernelGGL(tail_fixup_kernel<8>, g, b, 0, st, p);
  else hipLaunchKernelGGL(tail_fixup_kernel<16>, g, b, 0, st, p);
    DIC_LAUNCH_CHECK();
  }
  if (g_prof_on) {
    (void)hipEventRecord(rec.e1, st);
    g_prof_recs.push_back(rec);
  }
  if (p.splitk > 1 && !p.raw_partials) {
    const long long total = (long long)p.M * p.N;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 2048);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
    DIC_LAUNCH_CHECK();
  }
  return DIC_OK;
}

// n <= kGemmGroupMax independent contractions, all with K-major ("colk") A and B operands and 64x64 tiles, in ONE launch (+ one
// launch that reduces the split-K slabs of those that asked for a K split; each needs its own `ws` region).  Same arithmetic and
// summation order per problem as gemm_launch(p, st, 64).
int gemm_launch_group_colk(GemmParams* ps, int n, hipStream_t st) {
  DIC_REQUIRE(n >= 1 && n <= kGemmGroupMax, "gemm group: 1..%d problems", kGemmGroupMax);
  GemmGroup g{}, r{};
  g.n = n;
  int blocks = 0, rblocks = 0;
  double flops = 0.0, bytes = 0.0;
  for (int i = 0; i < n; ++i) {
    GemmParams& p = ps[i];
    DIC_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0 && p.A.p && p.B.p && p.ep.C, "gemm group: bad problem %d", i);
    DIC_REQUIRE(p.A.kind == OPK_COLK && p.B.kind == OPK_COLK && !p.ep.stats && !p.raw_partials, "gemm group: K-major operands only");
    p.mtiles = ceil_div(p.M, 64); p.ntiles = ceil_div(p.N, 64);
    const int nk = ceil_div(p.K, BK);
    if (p.splitk < 1) p.splitk = 1;
    if (p.splitk > nk) p.splitk = nk;
    p.ktiles_per_split = ceil_div(nk, p.splitk);
    p.splitk = ceil_div(nk, p.ktiles_per_split);
    DIC_REQUIRE(p.splitk == 1 || p.ws != nullptr, "gemm group: split-K needs a workspace");
    if (p.ep.alpha == 0.0f) p.ep.alpha = 1.0f;
    p.tail_first_block = p.mtiles * p.ntiles * p.splitk; p.tail_first_tile = 0; p.tail_split = 1; p.tail_ws = nullptr;
    g.first[i] = blocks;
    blocks += p.mtiles * p.ntiles * p.splitk;
    g.p[i] = p;
    flops += 2.0 * p.M * p.N * (double)p.K;
    bytes += 4.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N);
    if (p.splitk > 1) {
      r.first[r.n] = rblocks;
      rblocks += (int)std::min<long long>(((long long)p.M * p.N + 255) / 256, 512);
      r.p[r.n++] = p;
    }
  }
  g.first[n] = blocks;
  r.first[r.n] = rblocks;
  ProfRec rec{};
  if (g_prof_on) {
    rec.e0 = prof_event(); rec.e1 = prof_event();
    rec.flops = flops; rec.bytes = bytes; rec.key = OPK_COLK * 10 + OPK_COLK;
    (void)hipEventRecord(rec.e0, st);
  }
  hipLaunchKernelGGL((gemm_group_kernel<64, 64, OPK_COLK, OPK_COLK>), dim3(blocks), dim3(256), 0, st, g);
  DIC_LAUNCH_CHECK();
  if (g_prof_on) {
    (void)hipEventRecord(rec.e1, st);
    g_prof_recs.push_back(rec);
  }
  if (r.n > 0) {
    hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3(rblocks), dim3(256), 0, st, r);
    DIC_LAUNCH_CHECK();
  }
  return DIC_OK;
}

}  // namespace dic
