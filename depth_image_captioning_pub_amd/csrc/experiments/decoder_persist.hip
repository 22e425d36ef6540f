// Persistent forward loop of the soft-attention decoder: ONE launch runs all T decode steps
// (CD_RNNDecoderWithSoftAttention.forward, Depth_caption_model/depth_models.py:179-201; Soft_Attention.forward,
// attention.py:81-95), replacing 2 dependent launches per step (attn_fwd_kernel + the skinny gate GEMM).
//
// Decomposition: 256 workgroups = 16 row groups x 16 channel chunks, one per CU.  Workgroup (g, c) owns RPG = 4 batch
// rows and the 128 feature channels [c*128, +128) for the whole sequence:
//   * its slice of the fused feature map F[rows, :, chunk] stays in LDS for all T steps when it fits (the compact 49-cell
//     layout: 4 x 49 x 128 floats = 98 KB) - the per-step HBM/L2 pass over F disappears; with 196 cells it is streamed;
//   * its slices of W_h (attention query) and W_beta (gate) live in registers (32 + 32 floats per thread);
//   * per step it recomputes the cheap row-wide quantities (LSTM cell, q, the L scores, softmax) for its 4 rows, computes
//     context / gate / LSTM input for its channel chunk, and the PARTIAL gate pre-activations over its K slice
//     (128 context channels + 8 of the 128 recurrent inputs): slab[g][c][r][0:512];
//   * the 16 chunk workgroups of a row group exchange those slabs through global memory once per step: sc1
//     (write-through) stores, drained, one agent-scope atomic add per workgroup on the group's monotonic counter; the
//     readers poll it with sc1 loads, pass a workgroup barrier and read the slabs with sc1 loads
//     (MI355X_MICROARCH.md, inter-workgroup visibility, third valid form; 16-byte sc1 buffer stores / loads).
//   * placement (speed only; correct under any): blocks b and b+8 share an XCD, and the two channel chunks 2x, 2x+1 of
//     ALL row groups are dealt to XCD x, so the slice of [W_ih | W_hh] an XCD needs every step is 0.5 MB and stays in its
//     4-MB L2 (with a row group per XCD every XCD re-streamed the whole 4.7 MB through its fabric port each step:
//     7.8 us of the 20-us step).  The slabs are write-through either way, so their exchange does not care.
//   * the embedding part of the gate pre-activation is time-invariant under teacher forcing: Gemb = emb W_ih[:, :E]^T + b
//     is one GEMM before the loop.
// Every spin is bounded (wall clock): on time-out the workgroup raises the status word, poisons its rows' outputs with NaN
// and leaves, so the grid always drains and a lost hand-off can never be mistaken for a result.
#include "decoder.h"

namespace dic {

constexpr int kRPG = 4;                   // batch rows per row group
constexpr int kNG = 16, kNCk = 16;        // row groups, channel chunks (kNG * kNCk = 256 workgroups)
constexpr int kCW = kD / kNCk;            // 128 channels per chunk
constexpr int kHK = kH / kNCk;            // 8 recurrent inputs of the gate GEMM per chunk
static_assert(kCW == 128 && kH == 128 && kA == 128 && kG == 512, "persistent decoder: thread mappings assume these sizes");

__device__ __forceinline__ float ld_sc1(const float* p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned int*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_sc1(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned int*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct PersistFwdArgs {
  const float *F, *P;                      // fused features [B,L,D], hoisted P = W_z F + b_z [B,L,A]
  const float *WhT, *b_h, *w_full, *b_full, *WbT, *b_beta, *WcatT;
  const float* Gemb;                       // [B*T, 4H]: embedding part of the gate pre-activations + (b_ih + b_hh)
  const float* drop;                       // [B,T,H] or null
  const int* dec_len;                      // device copy
  float *Hall, *Call, *Gact, *Hdrop, *Qall, *ctx_all, *gate_all, *Xall, *alphas;
  float* slab;                             // [2][kNG][kNCk][kRPG][4H]
  unsigned int* sync;                      // [kNG] arrival counters, [kNG] = status (0 ok, 1 = a spin timed out)
  const int* packed_off;                   // device [T+1]: packed row offset of step t (host plan)
  int B, T;
  unsigned long long timeout_ticks;
  int placement;
  unsigned long long* dbg;                 // optional [T][8] phase time stamps of workgroup (g 0, c 0) (100 MHz ticks)
};
#define DBG_STAMP(i) do { if (a.dbg && bid == 0 && tid == 0 && t < a.T) a.dbg[t * 8 + (i)] = wall_clock64(); } while (0)

// wait until the group's counter reaches `target`; one lane polls, everybody passes the barrier.  false = timed out.
__device__ __forceinline__ bool group_wait(unsigned int* cnt, unsigned int target, unsigned long long timeout, int* flag_s) {
  if (threadIdx.x == 0) {
    const unsigned long long t0 = wall_clock64();
    int ok = 1;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > timeout) { ok = 0; break; }
    }
    *flag_s = ok;
  }
  __syncthreads();
  return *flag_s != 0;
}

template <int L, bool F_IN_LDS>
__global__ void __launch_bounds__(512, 2) decoder_fwd_persistent_kernel(const PersistFwdArgs a) {
  extern __shared__ __align__(16) float lds[];
  float* F_s = lds;                                              // [kRPG][L][kCW] when F_IN_LDS
  float* base = lds + (F_IN_LDS ? kRPG * L * kCW : 0);
  float* h_s = base;                                             // [kRPG][kH]
  float* q_s = h_s + kRPG * kH;                                  // [kRPG][kA]
  float* x_s = q_s + kRPG * kA;                                  // [kRPG][kCW]
  float* al_s = x_s + kRPG * kCW;                                // [kRPG][L padded to 200]
  float* part_s = al_s + kRPG * 200;                             // 32 KB scratch, one use at a time: [4][kRPG][128] q / gate
                                                                 // partials; [kRPG][4H] gate pre-activations (phase 1);
                                                                 // [4][kRPG][4H] K-quarter partials of phase 6
  __shared__ int flag_s;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int bid = blockIdx.x, slot = bid >> 3;
  // placement (speed only): 0 = the 16 chunk workgroups of a row group on one XCD; 1 = chunks 2x, 2x+1 of all groups on XCD x
  const int c = a.placement ? 2 * (bid & 7) + (slot & 1) : (slot & 15);
  const int g = a.placement ? (slot >> 1) : (bid & 7) + 8 * (slot >> 4);
  const int r0 = g * kRPG;
  if (r0 >= a.B) return;
  const int T = a.T;
  const int rmap = tid >> 7, cmap = tid & 127;                   // (row, column) mapping of the 4 x 128 element passes
  const int brow = r0 + rmap;                                    // batch row of this thread in those passes
  const bool row_ok = brow < a.B;
  const int len_row = row_ok ? a.dec_len[brow] : 0;
  const int len_grp = a.dec_len[r0];                             // lengths are sorted descending: the group's longest
  unsigned int* cnt = a.sync + g;

  // ---- one-time loads: weight slices into registers, the feature slice into LDS -------------------------------
  float wq[32], wb[32];
  {
    const int kq = tid >> 7;
#pragma unroll
    for (int k = 0; k < 32; ++k) wq[k] = a.WhT[(kq * 32 + k) * kA + cmap];
#pragma unroll
    for (int k = 0; k < 32; ++k) wb[k] = a.WbT[(long long)(kq * 32 + k) * kD + c * kCW + cmap];
  }
  if (F_IN_LDS) {
    for (int i = tid; i < kRPG * L * (kCW / 4); i += 512) {
      const int d4 = i % (kCW / 4), rl = i / (kCW / 4);
      const int r = rl / L, l = rl % L;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r0 + r < a.B) v = *reinterpret_cast<const float4*>(a.F + ((long long)(r0 + r) * L + l) * kD + c * kCW + d4 * 4);
      *reinterpret_cast<float4*>(F_s + (r * L + l) * kCW + d4 * 4) = v;
    }
  }
  const auto slab_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.slab, 0, (int)(2u * kNG * kNCk * kRPG * kG * sizeof(float)), 0x00020000);
  const float bq = a.b_h[cmap], bbeta = a.b_beta[c * kCW + cmap], bfull = a.b_full[0];
  float c_state = row_ok ? a.Call[((long long)brow * (T + 1)) * kH + cmap] : 0.f;      // c_0
  h_s[rmap * kH + cmap] = row_ok ? a.Hall[((long long)brow * (T + 1)) * kH + cmap] : 0.f;   // h_0
  bool dead = false;
  __syncthreads();

  for (int t = 0; t <= len_grp; ++t) {
    // ---- phase 1 (t > 0): LSTM cell of step t-1 from the group's partial slabs -> h_t, c_t --------------------
    DBG_STAMP(0);
    if (t > 0) {
      if (!group_wait(cnt, (unsigned)(kNCk * t), a.timeout_ticks, &flag_s)) { dead = true; break; }
      DBG_STAMP(1);
      const int tp = t - 1;
      {   // sum of the 16 chunk partials of gate columns 4*cmap..+3 of row rmap: 16 x 16-B sc1 loads in flight
        typedef float f4 __attribute__((ext_vector_type(4)));
        const unsigned off0 = (unsigned)((((tp & 1) * kNG + g) * kNCk) * (kRPG * kG) + rmap * kG + cmap * 4) * 4u;
        f4 v[kNCk];
#pragma unroll
        for (int z = 0; z < kNCk; ++z)
          v[z] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(slab_rsrc, off0 + (unsigned)z * (kRPG * kG * 4u), 0, 16));
        f4 sacc = v[0];
#pragma unroll
        for (int z = 1; z < kNCk; ++z) sacc += v[z];
        *reinterpret_cast<f4*>(part_s + rmap * kG + cmap * 4) = sacc;
      }
      __syncthreads();
      if (tp < len_row) {
        const float* ge = a.Gemb + ((long long)brow * T + tp) * kG + cmap;
        float pre[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) pre[q] = part_s[rmap * kG + q * kH + cmap] + ge[q * kH];
        const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
        c_state = fg * c_state + ig * gg;
        const float h = og * tanhf(c_state);
        h_s[rmap * kH + cmap] = h;
        if (c == 0) {                     // the tape (read by the backward and the vocabulary projection)
          const long long hc = ((long long)brow * (T + 1) + t) * kH + cmap;
          a.Call[hc] = c_state;
          a.Hall[hc] = h;
          float* ga = a.Gact + ((long long)brow * T + tp) * kG;
          ga[cmap] = ig; ga[kH + cmap] = fg; ga[2 * kH + cmap] = gg; ga[3 * kH + cmap] = og;
          const float dm = a.drop ? a.drop[((long long)brow * T + tp) * kH + cmap] : 1.0f;
          a.Hdrop[((long long)a.packed_off[tp] + brow) * kH + cmap] = h * dm;
        }
      }
      __syncthreads();
    }
    if (t == len_grp) break;
    DBG_STAMP(2);
    const bool act = t < len_row;           // this thread's row takes part in step t
    const long long bt = (long long)brow * T + t;

    // ---- phase 2: q = W_h h + b_h (four K quarters per output, weights in registers) ---------------------------
    {
      const int kq = tid >> 7;
#pragma unroll
      for (int r = 0; r < kRPG; ++r) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 32; k += 4) {
          const float4 h4 = *reinterpret_cast<const float4*>(h_s + r * kH + kq * 32 + k);
          s += wq[k] * h4.x + wq[k + 1] * h4.y + wq[k + 2] * h4.z + wq[k + 3] * h4.w;
        }
        part_s[(kq * kRPG + r) * 128 + cmap] = s;
      }
    }
    __syncthreads();
    {
      const float q = bq + ((part_s[(0 * kRPG + rmap) * 128 + cmap] + part_s[(1 * kRPG + rmap) * 128 + cmap]) +
                            (part_s[(2 * kRPG + rmap) * 128 + cmap] + part_s[(3 * kRPG + rmap) * 128 + cmap]));
      q_s[rmap * kA + cmap] = q;
      if (c == 0 && act) {
        a.Qall[bt * kA + cmap] = q;
        a.Xall[bt * kXK + kE + kD + cmap] = h_s[rmap * kH + cmap];       // h_{t-1} slot of the LSTM input (for dW_hh)
      }
    }
    __syncthreads();

    DBG_STAMP(3);
    // ---- phase 3: scores e[r][l] = w . relu(P[r,l,:] + q[r]) + b: one half-wave per (row, cell) pair -----------
    {
      const int l32 = lane & 31, hw = wv * 2 + (lane >> 5);
      const float4 w4 = *reinterpret_cast<const float4*>(a.w_full + l32 * 4);
      constexpr int NP = kRPG * L, NPASS = (NP + 15) / 16, BATCH = 13;
      for (int p0 = 0; p0 < NPASS; p0 += BATCH) {
        float4 p4[BATCH];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {            // branch-free guard: pairs past the end re-read a valid pair
          const int pr = min(hw + 16 * (p0 + i), NP - 1);
          const int r = pr / L, l = pr - r * L;
          const int bb = min(r0 + r, a.B - 1);
          p4[i] = *reinterpret_cast<const float4*>(a.P + ((long long)bb * L + l) * kA + l32 * 4);
        }
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
          const int pr = hw + 16 * (p0 + i);
          const int r = min(pr, NP - 1) / L;
          const float4 q4 = *reinterpret_cast<const float4*>(q_s + r * kA + l32 * 4);
          float sc = w4.x * fmaxf(p4[i].x + q4.x, 0.f) + w4.y * fmaxf(p4[i].y + q4.y, 0.f) +
                     w4.z * fmaxf(p4[i].z + q4.z, 0.f) + w4.w * fmaxf(p4[i].w + q4.w, 0.f);
          sc = half_wave_sum(sc);
          if (pr < NP && p0 + i < NPASS && l32 == 0) al_s[r * 200 + (pr - r * L)] = sc + bfull;
        }
      }
    }
    __syncthreads();
    DBG_STAMP(4);
    // ---- phase 4: softmax over the L cells, wave r handles row r ----------------------------------------------
    if (wv < kRPG) {
      float z[(L + 63) / 64];
      float m = -INFINITY;
#pragma unroll
      for (int i = 0; i < (L + 63) / 64; ++i) {
        const int l = lane + 64 * i;
        z[i] = l < L ? al_s[wv * 200 + l] : -INFINITY;
        m = fmaxf(m, z[i]);
      }
      m = wave_max(m);
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < (L + 63) / 64; ++i) { z[i] = (lane + 64 * i < L) ? expf(z[i] - m) : 0.f; s += z[i]; }
      s = wave_sum(s);
      const bool wact = (r0 + wv < a.B) && t < a.dec_len[min(r0 + wv, a.B - 1)];
#pragma unroll
      for (int i = 0; i < (L + 63) / 64; ++i) {
        const int l = lane + 64 * i;
        if (l < L) {
          const float al = z[i] / s;
          al_s[wv * 200 + l] = al;
          if (c == 0 && wact) a.alphas[((long long)(r0 + wv) * T + t) * L + l] = al;
        }
      }
    }
    // gate pre-activation partials (weights in registers): W_beta[chunk] h, four K quarters
    {
      const int kq = tid >> 7;
#pragma unroll
      for (int r = 0; r < kRPG; ++r) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 32; k += 4) {
          const float4 h4 = *reinterpret_cast<const float4*>(h_s + r * kH + kq * 32 + k);
          s += wb[k] * h4.x + wb[k + 1] * h4.y + wb[k + 2] * h4.z + wb[k + 3] * h4.w;
        }
        part_s[(kq * kRPG + r) * 128 + cmap] = s;
      }
    }
    __syncthreads();
    DBG_STAMP(5);
    // ---- phase 5: context + gate + LSTM input for channel (chunk, cmap) of row rmap ---------------------------
    {
      float cx = 0.f;
      if (F_IN_LDS) {
        const float* fr = F_s + rmap * L * kCW + cmap;
        const float* ar = al_s + rmap * 200;
#pragma unroll 7
        for (int l = 0; l < L; ++l) cx += ar[l] * fr[l * kCW];
      } else if (row_ok) {
        const float* fr = a.F + (long long)brow * L * kD + c * kCW + cmap;
        const float* ar = al_s + rmap * 200;
        for (int l0 = 0; l0 < L; l0 += 14) {            // 14 loads in flight (L = 196 = 14 * 14; 49 = 3 * 14 + 7)
          float fv[14];
#pragma unroll
          for (int i = 0; i < 14; ++i) fv[i] = fr[(long long)min(l0 + i, L - 1) * kD];
#pragma unroll
          for (int i = 0; i < 14; ++i) cx += (l0 + i < L ? ar[l0 + i] : 0.f) * fv[i];
        }
      }
      const float gpre = bbeta + ((part_s[(0 * kRPG + rmap) * 128 + cmap] + part_s[(1 * kRPG + rmap) * 128 + cmap]) +
                                  (part_s[(2 * kRPG + rmap) * 128 + cmap] + part_s[(3 * kRPG + rmap) * 128 + cmap]));
      const float gt = sigmoidf_(gpre);
      const float xv = gt * cx;
      x_s[rmap * kCW + cmap] = xv;
      if (act) {
        const int d = c * kCW + cmap;
        a.ctx_all[bt * kD + d] = cx;
        a.gate_all[bt * kD + d] = gt;
        a.Xall[bt * kXK + kE + d] = xv;
      }
    }
    __syncthreads();
    DBG_STAMP(6);
    // ---- phase 6: partial gate pre-activations over this workgroup's K slice ------------------------------------
    //   thread (kq, n4): K quarter kq (32 context channels + 2 recurrent inputs) x gate columns 4*n4..+3: 34 x 16-B loads
    //   of the transposed weight (coalesced over n4) in two batches, 4 rows x 4 columns accumulated in registers;
    //   the four K quarters are summed through LDS and the slab row goes out as 16-B write-through stores
    {
      typedef float f4 __attribute__((ext_vector_type(4)));
      const int kq = tid >> 7, n4 = tid & 127;
      const float* wt = a.WcatT + (long long)(kE + c * kCW + kq * 32) * kG + n4 * 4;
      f4 acc[kRPG];
#pragma unroll
      for (int r = 0; r < kRPG; ++r) acc[r] = (f4)(0.f);
#pragma unroll 1
      for (int k0 = 0; k0 < 32; k0 += 16) {
        f4 wv4[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) wv4[k] = *reinterpret_cast<const f4*>(wt + (long long)(k0 + k) * kG);
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
          for (int r = 0; r < kRPG; ++r) acc[r] += wv4[k] * x_s[r * kCW + kq * 32 + k0 + k];
      }
      {
        const float* wh = a.WcatT + (long long)(kE + kD + c * kHK + kq * 2) * kG + n4 * 4;
        const f4 w0 = *reinterpret_cast<const f4*>(wh), w1 = *reinterpret_cast<const f4*>(wh + kG);
#pragma unroll
        for (int r = 0; r < kRPG; ++r)
          acc[r] += w0 * h_s[r * kH + c * kHK + kq * 2] + w1 * h_s[r * kH + c * kHK + kq * 2 + 1];
      }
#pragma unroll
      for (int r = 0; r < kRPG; ++r) *reinterpret_cast<f4*>(part_s + ((kq * kRPG + r) * kG) + n4 * 4) = acc[r];
      __syncthreads();
      const f4 s4 = (*reinterpret_cast<const f4*>(part_s + ((0 * kRPG + rmap) * kG) + cmap * 4) +
                     *reinterpret_cast<const f4*>(part_s + ((1 * kRPG + rmap) * kG) + cmap * 4)) +
                    (*reinterpret_cast<const f4*>(part_s + ((2 * kRPG + rmap) * kG) + cmap * 4) +
                     *reinterpret_cast<const f4*>(part_s + ((3 * kRPG + rmap) * kG) + cmap * 4));
      const unsigned off = (unsigned)(((((t & 1) * kNG + g) * kNCk + c) * kRPG + rmap) * kG + cmap * 4) * 4u;
      typedef unsigned u4 __attribute__((ext_vector_type(4)));
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, s4), slab_rsrc, off, 0, 16);      // aux 16 = sc1
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    DBG_STAMP(7);
  }

  if (dead) {      // a hand-off timed out: make the failure visible (status word) and un-mistakable (NaN outputs)
    if (tid == 0) __hip_atomic_store(a.sync + kNG, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (row_ok)
      for (int tt = 0; tt < len_row; ++tt) a.Hdrop[((long long)a.packed_off[tt] + brow) * kH + cmap] = __builtin_nanf("");
  }
}

static unsigned long long* g_persist_dbg = nullptr;
static int g_persist_placement = 0;
void decoder_persist_debug_placement(int p) { g_persist_placement = p; }
void decoder_persist_debug_buffer(unsigned long long* p) { g_persist_dbg = p; }

size_t decoder_persist_lds_bytes(int cells, bool f_in_lds) {
  size_t fl = (size_t)kRPG * kH + kRPG * kA + kRPG * kCW + kRPG * 200 + 4 * kRPG * 128;
  fl += 4 * kRPG * kG - 4 * kRPG * 128;          // the scratch region is sized for the phase-6 partials: [4][kRPG][4H]
  if (f_in_lds) fl += (size_t)kRPG * cells * kCW;
  return fl * sizeof(float);
}

bool decoder_persist_eligible(int B, int T, int mode) { return mode == 0 && B <= kNG * kRPG && T >= 1; }

int decoder_fwd_persistent(const DecoderWs& ws, const dic_decoder_weights* w, int B, int T, int cells, const float* drop_mult,
                           float* alphas, const int* host_packed_off, hipStream_t st) {
  PersistFwdArgs a{};
  a.F = ws.F; a.P = ws.P; a.WhT = ws.WhT; a.b_h = w->dec_att_b; a.w_full = w->full_att_w; a.b_full = w->full_att_b;
  a.WbT = ws.WbT; a.b_beta = w->fbeta_b; a.WcatT = ws.WcatT; a.Gemb = ws.Gemb; a.drop = drop_mult; a.dec_len = ws.dlen;
  a.Hall = ws.Hall; a.Call = ws.Call; a.Gact = ws.Gact; a.Hdrop = ws.Hdrop; a.Qall = ws.Qall; a.ctx_all = ws.ctx;
  a.gate_all = ws.gate; a.Xall = ws.Xall; a.alphas = alphas; a.slab = ws.pslab; a.sync = ws.psync;
  a.packed_off = ws.poff; a.B = B; a.T = T;
  a.timeout_ticks = 200000000ull;                 // 2 s of the 100 MHz wall clock
  a.dbg = g_persist_dbg;
  a.placement = g_persist_placement;
  DIC_CHECK_HIP(hipMemsetAsync(ws.psync, 0, sizeof(unsigned int) * 32, st));
  DIC_CHECK_HIP(hipMemcpyAsync(ws.poff, host_packed_off, sizeof(int) * (T + 1), hipMemcpyHostToDevice, st));
  const bool in_lds = cells == kLc;
  const size_t lds = decoder_persist_lds_bytes(cells, in_lds);
  if (cells == kLc) {
    static bool once = false;
    if (!once) {
      DIC_CHECK_HIP(hipFuncSetAttribute((const void*)decoder_fwd_persistent_kernel<kLc, true>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      once = true;
    }
    hipLaunchKernelGGL((decoder_fwd_persistent_kernel<kLc, true>), dim3(kNG * kNCk), dim3(512), lds, st, a);
  } else {
    hipLaunchKernelGGL((decoder_fwd_persistent_kernel<kL, false>), dim3(kNG * kNCk), dim3(512), lds, st, a);
  }
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

}  // namespace dic
