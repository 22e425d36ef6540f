// Decoder of the depth-soft / depth-hard captioner on MI355X.
//
// Replaces CD_RNNDecoderWith{Soft,Hard}Attention.forward/eval_forward/batch_sample
// (Captioning_models/Depth_caption_model/depth_models.py:153-305, 580-789) and
// Soft_Attention / Hard_Attention / Gumbel_softmax (Captioning_models/attention.py:12-167),
// plus their autograd backward (depth_train.py:219).
//
// Structure: time-invariant work is hoisted out of the T-step loop
// (P = Wz F + bz, quirk Q4; embedding half of the LSTM input; vocabulary projection and every
// weight gradient are batched over all steps as single MFMA GEMMs); per step only the
// HBM-bound attention/context kernels and one skinny fused LSTM GEMM run.
#include "decoder.h"
#include <algorithm>
#include <vector>

namespace dic {

// ------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------
DecoderWs decoder_carve(void* ws, size_t ws_bytes, int B, int T, int V, int N, bool* overflow) {
  Carver c(ws, ws_bytes);
  DecoderWs w{};
  const size_t BL = (size_t)B * kL, BT = (size_t)B * T;
  w.F = c.take<float>(BL * kD);
  w.P = c.take<float>(BL * kA);
  w.mean = c.take<float>((size_t)B * kD);
  w.Wcat = c.take<float>((size_t)kG * kXK);
  w.WcatT = c.take<float>((size_t)kG * kXK);
  w.bcat = c.take<float>(kG);
  w.WhT = c.take<float>((size_t)kH * kA);
  w.WbT = c.take<float>((size_t)kH * kD);
  w.WzT = c.take<float>((size_t)kA * kD);
  w.Xall = c.take<float>(BT * kXK);
  w.Hall = c.take<float>((size_t)B * (T + 1) * kH);
  w.Call = c.take<float>((size_t)B * (T + 1) * kH);
  w.Gact = c.take<float>(BT * kG);
  w.Qall = c.take<float>(BT * kA);
  w.ctx = c.take<float>(BT * kD);
  w.gate = c.take<float>(BT * kD);
  w.Hdrop = c.take<float>((size_t)N * kH);
  w.slab_g = c.take<float>((size_t)kS_LSTM * B * kG);
  size_t g = (size_t)16 * B * 2 * kH;                       // init_linear split-K
  g = std::max(g, (size_t)8 * N * kH);                      // dHd = dlogits * W_o   split-K 8
  g = std::max(g, (size_t)8 * kA * (kD + kH));              // dW_z and dW_q split-K 8, side by side (one grouped launch)
  g = std::max(g, (size_t)8 * B * kD);                      // dmean split-K
  w.gemm_ws_floats = g;
  w.gemm_ws = c.take<float>(g);
  w.dHd = c.take<float>((size_t)N * kH);
  w.slab_dx = c.take<float>((size_t)kS_DX * B * kXK);
  w.dG = c.take<float>(BT * kG);           // dG .. dq are adjacent: the backward zeroes them with one memset
  w.dctx = c.take<float>(BT * kD);
  w.dgpre = c.take<float>(BT * kD);
  w.dq = c.take<float>(BT * kA);
  w.dalp = c.take<float>((size_t)kNCH * B * kL);
  w.pbeta = c.take<float>((size_t)kNCH * B * kH);
  w.dqp = c.take<float>((size_t)kLCH * B * kA);
  w.dwf_acc = c.take<float>((size_t)kLCH * B * kA);
  w.dbf_acc = c.take<float>((size_t)kLCH * B);
  w.dPacc = c.take<float>(BL * kA);
  w.carry_dc = c.take<float>((size_t)B * kH);
  w.dinit = c.take<float>((size_t)B * 2 * kH);
  w.dmean = c.take<float>((size_t)B * kD);
  // one column sum at a time (V or kXK wide) or the batch of seven bias gradients (sum of their widths, <= 64 rows each)
  w.colsum_ws = c.take<float>((size_t)64 * std::max(std::max(V, kXK), kG + kD + 3 * kA + 2 * kH + 64));
  w.alpha_c = c.take<float>(BT * kLc);
  w.dalpha_c = c.take<float>(BT * kLc);
  w.dXe = c.take<float>(BT * kE);
  w.dlen = c.take<int>((size_t)B);
#ifdef DIC_EXPERIMENTS
  w.Gemb = c.take<float>(BT * kG);
  w.pslab = c.take<float>((size_t)2 * 16 * 16 * 4 * kG);
  w.psync = c.take<unsigned int>(32);
  w.poff = c.take<int>((size_t)T + 2);
#endif
  w.logits_step = c.take<float>((size_t)B * V);
  w.ids = c.take<long long>((size_t)B);
  w.bytes = c.off;
  if (overflow) *overflow = c.overflow;
  return w;
}

// ------------------------------------------------------------------------------------------
// small utility kernels
// ------------------------------------------------------------------------------------------
// out[c*R + r] = in[r*C + c]
__global__ void __launch_bounds__(256) transpose_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                         int R, int Cc) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int r = by + j, c = bx + tx;
    tile[j][tx] = (r < R && c < Cc) ? in[(long long)r * Cc + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = bx + j, r = by + tx;
    if (r < R && c < Cc) out[(long long)c * R + r] = tile[tx][j];
  }
}

// Wcat[g][0:2176] = W_ih[g][:], Wcat[g][2176:2304] = W_hh[g][:], bcat = b_ih + b_hh
__global__ void __launch_bounds__(256) pack_lstm_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                                         const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                         float* __restrict__ Wcat, float* __restrict__ bcat) {
  const int g = blockIdx.x;
  for (int k = threadIdx.x; k < kXK; k += 256)
    Wcat[(long long)g * kXK + k] = (k < kE + kD) ? w_ih[(long long)g * (kE + kD) + k] : w_hh[g * kH + (k - kE - kD)];
  if (threadIdx.x == 0) bcat[g] = b_ih[g] + b_hh[g];
}

// column sums of X[M][N] (row stride ld): stage 1 writes partial[RS][N]; stage 2 (RS rows) writes out[N]
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ X, long long ld, int M, int N,
                                                      float* __restrict__ out, int rs) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
#pragma unroll 8
  for (int m = blockIdx.y; m < M; m += rs) s += X[(long long)m * ld + n];
  out[(long long)blockIdx.y * N + n] = s;
}

// Several independent column sums in two launches (the seven bias gradients after BPTT were 14 dependent ~5-us launches).
// Per job the arithmetic is exactly colsum()'s: `rs` strided partial rows, then their sum in order.
struct ColsumJob { const float* X; long long ld; int M, N, rs; float* out; float* part; };
struct ColsumBatch { ColsumJob j[8]; };
__global__ void __launch_bounds__(256) colsum_batch_kernel(const ColsumBatch b, int stage) {
  const ColsumJob job = b.j[blockIdx.z];
  const int n = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (n >= job.N) return;
  if (stage == 1) {
    if (y >= job.rs) return;
    float s = 0.f;
#pragma unroll 8
    for (int m = y; m < job.M; m += job.rs) s += job.X[(long long)m * job.ld + n];
    (job.rs > 1 ? job.part : job.out)[(long long)y * job.N + n] = s;
  } else {
    if (job.rs == 1 || y != 0) return;
    float s = 0.f;
#pragma unroll 8
    for (int m = 0; m < job.rs; ++m) s += job.part[(long long)m * job.N + n];
    job.out[n] = s;
  }
}

static int colsum_batch(ColsumBatch& b, int njobs, float* ws, hipStream_t st) {
  int maxn = 1, maxrs = 1;
  float* part = ws;
  for (int i = 0; i < njobs; ++i) {
    ColsumJob& j = b.j[i];
    j.rs = std::min(64, std::max(1, j.M / 8));
    j.part = part;
    part += (size_t)j.rs * j.N;
    maxn = std::max(maxn, j.N); maxrs = std::max(maxrs, j.rs);
  }
  hipLaunchKernelGGL(colsum_batch_kernel, dim3(ceil_div(maxn, 256), maxrs, njobs), dim3(256), 0, st, b, 1);
  if (maxrs > 1) hipLaunchKernelGGL(colsum_batch_kernel, dim3(ceil_div(maxn, 256), 1, njobs), dim3(256), 0, st, b, 2);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

static int colsum(const float* X, long long ld, int M, int N, float* out, float* ws, hipStream_t st) {
  const int rs = std::min(64, std::max(1, M / 8));
  if (rs > 1) {
    hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(N, 256), rs), dim3(256), 0, st, X, ld, M, N, ws, rs);
    hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(N, 256), 1), dim3(256), 0, st, ws, (long long)N, rs, N, out, 1);
  } else {
    hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(N, 256), 1), dim3(256), 0, st, X, ld, M, N, out, 1);
  }
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

// F = F_rgb + F_depth ; mean[b,d] = sum_l F[b,l,d] / L          (depth_models.py:163,166)
// grid (D/256, B), 256 threads: wave w takes l = w, w+4, ...; lanes hold float4 over 256 channels
template <int L>
__global__ void __launch_bounds__(256) fuse_mean_kernel(const float* __restrict__ frgb, const float* __restrict__ fdep,
                                                         float* __restrict__ F, float* __restrict__ mean) {
  __shared__ float4 red[4][64];
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long base = (long long)b * L * kD + blockIdx.x * 256 + lane * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int l = w; l < L; l += 4) {
    const long long o = base + (long long)l * kD;
    float4 v = *reinterpret_cast<const float4*>(frgb + o);
    if (fdep) {
      const float4 u = *reinterpret_cast<const float4*>(fdep + o);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    *reinterpret_cast<float4*>(F + o) = v;
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  red[w][lane] = acc;
  __syncthreads();
  if (w == 0) {
    float4 s = red[0][lane];
#pragma unroll
    for (int i = 1; i < 4; ++i) { s.x += red[i][lane].x; s.y += red[i][lane].y; s.z += red[i][lane].z; s.w += red[i][lane].w; }
    const float inv = (float)L;
    s.x /= inv; s.y /= inv; s.z /= inv; s.w /= inv;
    *reinterpret_cast<float4*>(mean + (long long)b * kD + blockIdx.x * 256 + lane * 4) = s;
  }
}

// Xall[(b*T+t), 0:E] = embed[captions[b,t]]   for t < dec_len[b]        (depth_models.py:160,192)
__global__ void __launch_bounds__(128) embed_gather_kernel(const float* __restrict__ embed,
                                                            const long long* __restrict__ cap, int cap_stride,
                                                            const int* __restrict__ dec_len, int T, int V,
                                                            float* __restrict__ Xall) {
  const int b = blockIdx.y, t = blockIdx.x;
  if (t >= dec_len[b]) return;
  long long id = cap[(long long)b * cap_stride + t];
  id = id < 0 ? 0 : (id >= V ? V - 1 : id);
  Xall[((long long)b * T + t) * kXK + threadIdx.x] = embed[id * kE + threadIdx.x];
}

// ------------------------------------------------------------------------------------------
// forward step kernel 1: attention scores -> softmax / Gumbel -> context -> beta gate
//   (attention.py:84-93, 12-25, 40-46; depth_models.py:185-192)
// grid (kNCH, nb): workgroup (chunk, b) owns channels [chunk*256, +256) of batch row b.
// The score / softmax part (P[b]: 100 KB, L2-resident) is recomputed by the 8 chunk workgroups of a
// row; the HBM-heavy part - one pass over F[b,:,chunk] - is split between them.
// mode 0: softmax(e); 1: softmax((e+g)/temp); 2: one-hot(argmax(e+g)), g = -log(-log(u)).
// ------------------------------------------------------------------------------------------
// LSTM pointwise of the PREVIOUS step, fused into the prologue of attn_fwd (saves one dependent launch per decode
// step): every chunk-workgroup of row b recomputes h_t from the gate-GEMM slabs of step t-1 (72 loads per thread, one
// round trip); the chunk-0 workgroup also stores h_t / c_t / the gate activations / the dropped hidden state, exactly
// what lstm_fwd_kernel(t-1) would have.  Rows that were active at t-1 but not at t get only that part.
struct FusedLstm {
  const float* slab;        // [nslab][nb_prev][4H] partial gate pre-activations of step t-1 (null: not fused)
  const float* bcat;
  const float* drop;        // dropout multiplier [B][T][H] or null
  float *Hall_w, *Call, *Gact, *Hdrop;
  int nslab, nb_prev, nb_cur, packed_off_prev;
};

#ifdef DIC_EXPERIMENTS
// phase time stamps of attn_fwd_kernel (s_memtime, workgroup 0 / a middle workgroup, thread 0): scripts/diag_attn_phases.py
__device__ unsigned long long g_attn_stamps[2][16];
#define DIC_ATTN_STAMP(I_)                                                                                   \
  if (threadIdx.x == 0 && (lin == 0 || lin == 257)) g_attn_stamps[lin == 0 ? 0 : 1][I_] = __builtin_amdgcn_s_memtime();
#else
#define DIC_ATTN_STAMP(I_)
#endif

template <int L>
__global__ void __launch_bounds__(512, 4) attn_fwd_kernel(
    const float* __restrict__ F, const float* __restrict__ P, const float* __restrict__ Hall,
    const float* __restrict__ WhT, const float* __restrict__ b_h, const float* __restrict__ w_full,
    const float* __restrict__ b_full, const float* __restrict__ WbT, const float* __restrict__ b_beta,
    int t, int T, int mode, const float* __restrict__ gumbel_u, int B, float temp,
    float* __restrict__ alphas, float* __restrict__ Qall, float* __restrict__ ctx_all,
    float* __restrict__ gate_all, float* __restrict__ Xall, int do_gate, const FusedLstm fl, const int nrows) {
  // A step is a chain of short phases; what it costs is memory round trips, not bytes.  So every load whose ADDRESS does not
  // depend on the previous phase is issued as early as registers allow (round 3): the W_h slice goes out together with the
  // LSTM slabs at the very top, the P rows under the q phase; the W_beta slice (its product needs only h) streams in two
  // batches under the q / score / softmax phases; the F rows (whose weights, not addresses, come from the softmax) are in
  // flight while the softmax runs (compact layout: all of them; 196 cells: the first half).  Per-element arithmetic and summation orders are those of the round-1 kernel.
  __shared__ float h_s[kH];
  __shared__ float q_s[4][kA];
  constexpr int NPS = (L + 15) / 16;                 // score passes: 16 cells (half-waves) per pass
  constexpr int NB = ((L + 7) / 8 + 1) / 2;          // context loop: two batches of NB cells per wave (8 waves)
  constexpr int EP = 16 * NB;                        // padded cell count of the context loop
  constexpr bool kCompact = L <= 64;                 // 49 cells: P rows and BOTH F batches fit in registers ahead of their use
  __shared__ float e_s[EP];
  __shared__ float red_s[16];
  __shared__ __align__(16) float cred[8][256];
  __shared__ float gp_s[2][256];
  // Workgroup (chunk, b) owns channels [chunk*256, +256) of batch row b.  The eight chunk workgroups of a row share its P
  // rows and LSTM slabs (each recomputes scores and cell): they are mapped to dispatch ids with equal id % 8, i.e. to ONE
  // XCD under round-robin placement (speed only), so those bytes leave HBM / the Infinity Cache once per row, not eight times.
  const int lin = blockIdx.y * kNCH + blockIdx.x;
  const int b = (lin & 7) + 8 * (lin >> 6), chunk = (lin >> 3) & 7;
  if (b >= nrows) return;
  const int tid = threadIdx.x, lane = tid & 63;
  // wave index as a scalar: every address below is (uniform base) + (small per-lane offset), which keeps the
  // many loads in flight from costing a 64-bit address register pair each
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long bt = (long long)b * T + t;
  const int l32 = lane & 31, hw = w * 2 + (lane >> 5);
  const bool lstm_only = fl.slab && b >= fl.nb_cur;              // row finished at t-1: only chunk 0 completes its last cell
  if (lstm_only && chunk != 0) return;
  DIC_ATTN_STAMP(0)

  // ---- early load (compact layout): q weights (W_h^T quarter of this wave), in flight together with the LSTM slabs
  float wv[32];
  float4 p4[NPS];
  const int quarter = w >> 1;
  const float* Pu = P + (long long)b * L * kA;                       // uniform
  const float* Wq = WhT + quarter * 32 * kA + (w & 1) * 64;
  if constexpr (kCompact) {
    if (!lstm_only) {
#pragma unroll
      for (int k = 0; k < 32; ++k) wv[k] = Wq[k * kA + lane];
    }
  }

  if (fl.slab) {               // h_t = LSTM cell of step t-1 (see FusedLstm)
    if (tid < kH) {
      const int j = tid, tp = t - 1;
      float pre[4];
      float v[4][kS_LSTM];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int z = 0; z < kS_LSTM; ++z)
          v[q][z] = (z < fl.nslab) ? fl.slab[((long long)z * fl.nb_prev + b) * kG + q * kH + j] : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float sacc = v[q][0];
#pragma unroll
        for (int z = 1; z < kS_LSTM; ++z) sacc += v[q][z];
        pre[q] = sacc + fl.bcat[q * kH + j];
      }
      const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
      const long long hc = ((long long)b * (T + 1) + tp) * kH + j;
      const float c = fg * fl.Call[hc] + ig * gg;
      const float h = og * tanhf(c);
      h_s[j] = h;
      if (chunk == 0) {
        fl.Call[hc + kH] = c;
        fl.Hall_w[hc + kH] = h;
        float* ga = fl.Gact + ((long long)b * T + tp) * kG;
        ga[j] = ig; ga[kH + j] = fg; ga[2 * kH + j] = gg; ga[3 * kH + j] = og;
        const float dm = fl.drop ? fl.drop[((long long)b * T + tp) * kH + j] : 1.0f;
        fl.Hdrop[((long long)fl.packed_off_prev + b) * kH + j] = h * dm;
      }
    }
    if (lstm_only) return;                                 // (whole workgroup: b is uniform)
  } else if (tid < kH) {
    h_s[tid] = Hall[((long long)b * (T + 1) + t) * kH + tid];
  }
  if (tid >= 256 && tid < 256 + (EP - L)) e_s[L + tid - 256] = 0.f;  // padding cells of the context loop
  __syncthreads();
  DIC_ATTN_STAMP(1)
  const int dl = tid & 255, half = w >> 2;
  const float* Wg = WbT + (long long)(half * 64) * kD + chunk * 256 + (w & 3) * 64;           // uniform
  const float* Fu = F + (long long)b * L * kD + chunk * 256;                              // uniform
  const unsigned foff = lane * 4;
  float wg[32];
  float4 v0[NB], v1[kCompact ? NB : 1];
  float gs = 0.f;
  if constexpr (kCompact) {
    // gate weights, first half of this wave's K range, and the P rows of the score phase: in flight under the q phase
    if (do_gate) {
#pragma unroll
      for (int k = 0; k < 32; ++k) wg[k] = Wg[(long long)k * kD + lane];
    }
#pragma unroll
    for (int i = 0; i < NPS; ++i) {     // branch-free guard: cells past the end re-read the last cell (never stored)
      const unsigned poff = (unsigned)min(hw + 16 * i, L - 1) * kA + l32 * 4;
      p4[i] = *reinterpret_cast<const float4*>(Pu + poff);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 32; ++k) wv[k] = Wq[k * kA + lane];
  }
  {  // q = Wh h + bh   (four quarters of K per output)
    const int a = tid & (kA - 1);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) s += wv[k] * h_s[quarter * 32 + k];
    q_s[quarter][a] = s;
  }
  __syncthreads();
  DIC_ATTN_STAMP(2)
  if (tid < kA) {
    const float q = b_h[tid] + ((q_s[0][tid] + q_s[1][tid]) + (q_s[2][tid] + q_s[3][tid]));
    q_s[0][tid] = q;
    if (chunk == 0 && Qall) Qall[bt * kA + tid] = q;
  }
  if (chunk == 0 && tid < kH && Xall) Xall[bt * kXK + kE + kD + tid] = h_s[tid];   // h_prev slot of the LSTM input
  __syncthreads();
  DIC_ATTN_STAMP(3)
  {  // e[l] = w . relu(P[l,:] + q) + b : one 32-lane half-wave per cell, float4 per lane, 16 cells per pass
    const float4 q4 = *reinterpret_cast<const float4*>(&q_s[0][l32 * 4]);
    const float4 w4 = *reinterpret_cast<const float4*>(w_full + l32 * 4);
    const float bf = b_full[0];
    if constexpr (!kCompact) {
#pragma unroll
      for (int i = 0; i < NPS; ++i) {
        const unsigned poff = (unsigned)min(hw + 16 * i, L - 1) * kA + l32 * 4;
        p4[i] = *reinterpret_cast<const float4*>(Pu + poff);
      }
    }
#pragma unroll
    for (int i = 0; i < NPS; ++i) {
      const int l = hw + 16 * i;
      float sc = w4.x * fmaxf(p4[i].x + q4.x, 0.f) + w4.y * fmaxf(p4[i].y + q4.y, 0.f) +
                 w4.z * fmaxf(p4[i].z + q4.z, 0.f) + w4.w * fmaxf(p4[i].w + q4.w, 0.f);
      sc = half_wave_sum(sc);
      if (l < L && l32 == 0) e_s[l] = sc + bf;
    }
  }
  if constexpr (kCompact) {
    // the one HBM pass of the step: F[b, :, chunk] (wave w takes cells w, w+8, ...), issued before the softmax that weighs it
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int l = min(w + 8 * i, L - 1);                 // padding cells re-read the last cell, weight e_s = 0
      v0[i] = *reinterpret_cast<const float4*>(Fu + (long long)l * kD + foff);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int l = min(w + 8 * (NB + i), L - 1);
      v1[i] = *reinterpret_cast<const float4*>(Fu + (long long)l * kD + foff);
    }
    if (do_gate) {             // gate pre-activation, first half of K; then the second half's weights go out
#pragma unroll
      for (int k = 0; k < 32; ++k) gs += wg[k] * h_s[half * 64 + k];
      __builtin_amdgcn_sched_barrier(0);         // (the second half re-uses the registers of the first: keep the order)
#pragma unroll
      for (int k = 0; k < 32; ++k) wg[k] = Wg[(long long)(32 + k) * kD + lane];
    }
  }
  DIC_ATTN_STAMP(4)
  __syncthreads();
  DIC_ATTN_STAMP(5)
  {  // attention weights over the L cells
    float z = -INFINITY;
    if (tid < L) {
      z = e_s[tid];
      if (mode != 0) {
        const float u = gumbel_u[((long long)t * B + b) * L + tid];
        z += -logf(-logf(u));
        if (mode == 1) z /= temp;
      }
    }
    float m = wave_max(z);
    if (lane == 0) red_s[w] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red_s[0], red_s[1]), fmaxf(red_s[2], red_s[3]));      // cells live in waves 0..3
    float al;
    if (mode == 2) {   // first index attaining the maximum -> one-hot
      int cand = (tid < L && z == m) ? tid : 0x7fffffff;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
      __syncthreads();
      if (lane == 0) red_s[8 + w] = __int_as_float(cand);
      __syncthreads();
      const int win = min(min(__float_as_int(red_s[8]), __float_as_int(red_s[9])),
                          min(__float_as_int(red_s[10]), __float_as_int(red_s[11])));
      al = (tid == win) ? 1.f : 0.f;
    } else {
      const float ex = (tid < L) ? expf(z - m) : 0.f;
      const float sm = wave_sum(ex);
      if (lane == 0) red_s[8 + w] = sm;
      __syncthreads();
      al = ex / (red_s[8] + red_s[9] + red_s[10] + red_s[11]);
    }
    __syncthreads();
    if (tid < L) {
      e_s[tid] = al;
      if (chunk == 0) alphas[bt * L + tid] = al;
    }
  }
  __syncthreads();
  DIC_ATTN_STAMP(6)
  {  // ctx[d] = sum_l alpha[l] F[b,l,d] over this chunk (two batches of NB cells per wave), fused with the pre-activation of
     // gate = sigmoid(W_beta h + b) for the same 256 channels (two halves of K per channel)
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (kCompact) {            // everything is in registers already
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const float a = e_s[w + 8 * i];                    // padded with zeros up to EP
        acc.x += a * v0[i].x; acc.y += a * v0[i].y; acc.z += a * v0[i].z; acc.w += a * v0[i].w;
      }
      if (do_gate) {
#pragma unroll
        for (int k = 0; k < 32; ++k) gs += wg[k] * h_s[half * 64 + 32 + k];
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const float a = e_s[w + 8 * (NB + i)];
        acc.x += a * v1[i].x; acc.y += a * v1[i].y; acc.z += a * v1[i].z; acc.w += a * v1[i].w;
      }
    } else {                             // 196 cells: two batches of 13 x 16 B + 32 x 4 B per lane, each one round trip
#pragma unroll 1
      for (int bt2 = 0; bt2 < 2; ++bt2) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const int l = min(w + 8 * (bt2 * NB + i), L - 1);
          v0[i] = *reinterpret_cast<const float4*>(Fu + (long long)l * kD + foff);
        }
        if (do_gate) {
#pragma unroll
          for (int k = 0; k < 32; ++k) wg[k] = Wg[(long long)(bt2 * 32 + k) * kD + lane];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const float a = e_s[w + 8 * (bt2 * NB + i)];
          acc.x += a * v0[i].x; acc.y += a * v0[i].y; acc.z += a * v0[i].z; acc.w += a * v0[i].w;
        }
        if (do_gate) {
#pragma unroll
          for (int k = 0; k < 32; ++k) gs += wg[k] * h_s[half * 64 + bt2 * 32 + k];
        }
      }
    }
    *reinterpret_cast<float4*>(&cred[w][lane * 4]) = acc;
    gp_s[half][dl] = gs;
  }
  DIC_ATTN_STAMP(7)
  __syncthreads();
  DIC_ATTN_STAMP(8)
  if (tid < 256) {  // x = gate * ctx          (depth_models.py:189-190)
    const int d = chunk * 256 + tid;
    const float c = ((cred[0][tid] + cred[1][tid]) + (cred[2][tid] + cred[3][tid])) +
                    ((cred[4][tid] + cred[5][tid]) + (cred[6][tid] + cred[7][tid]));
    ctx_all[bt * kD + d] = c;
    if (do_gate) {            // (stand-alone Soft/Hard_Attention.forward stops at the context vector)
      const float g = sigmoidf_(b_beta[d] + (gp_s[0][tid] + gp_s[1][tid]));
      gate_all[bt * kD + d] = g;
      Xall[bt * kXK + kE + d] = g * c;
    }
  }
  DIC_ATTN_STAMP(9)
}

// ------------------------------------------------------------------------------------------
// forward step kernel 3: LSTM cell pointwise (reduces the split-K slabs of the gate GEMM)
//   nn.LSTMCell gate order i,f,g,o (depth_models.py:193-194) + dropout on h for the vocabulary
//   projection (depth_models.py:197; the carried h is NOT dropped).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kH) lstm_fwd_kernel(const float* __restrict__ slab, int nslab, int nb,
                                                       const float* __restrict__ bcat, int t, int T,
                                                       const float* __restrict__ drop, int packed_off,
                                                       float* __restrict__ Hall, float* __restrict__ Call,
                                                       float* __restrict__ Gact, float* __restrict__ Hdrop) {
  const int b = blockIdx.x, j = threadIdx.x;
  float pre[4];
  {   // all 4 x nslab partials in flight at once (nslab <= kS_LSTM), summed in slab order
    float v[4][kS_LSTM];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int z = 0; z < kS_LSTM; ++z) v[q][z] = (z < nslab) ? slab[((long long)z * nb + b) * kG + q * kH + j] : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float sacc = v[q][0];
#pragma unroll
      for (int z = 1; z < kS_LSTM; ++z) sacc += v[q][z];
      pre[q] = sacc + bcat[q * kH + j];
    }
  }
  const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
  const long long hc = ((long long)b * (T + 1) + t) * kH + j;
  const float c = fg * Call[hc] + ig * gg;
  const float h = og * tanhf(c);
  Call[hc + kH] = c;
  Hall[hc + kH] = h;
  float* ga = Gact + ((long long)b * T + t) * kG;
  ga[j] = ig; ga[kH + j] = fg; ga[2 * kH + j] = gg; ga[3 * kH + j] = og;
  const float dm = drop ? drop[((long long)b * T + t) * kH + j] : 1.0f;
  Hdrop[((long long)packed_off + b) * kH + j] = h * dm;
}

// ------------------------------------------------------------------------------------------
// backward step kernel 1: assemble dh_t / dc_t and LSTM pointwise backward
//   dh_t = W_o^T dlogit_t (dropout mask applied)  +  carry from step t+1, where the carry is
//   assembled here from step t+1's products:  dX[:,h slot] + W_h^T dq + W_beta^T dgpre.
//   final=1: only assemble the carry into dinit (gradient of h0 | c0) after step 0.
// ------------------------------------------------------------------------------------------
// body: thread j of row b; `active` = this thread takes part (the fused kernel runs it on the first kH of 256 threads);
// every thread of the workgroup must call it (it contains a barrier)
__device__ __forceinline__ void lstm_bwd_body(
    const int b, const int j, const bool active,
    int t, int T, int B, int nb_next, int have_next, int final_pass, int nlch,
    const float* __restrict__ dHd, int packed_off, const float* __restrict__ drop,
    const float* __restrict__ slab_dx, int nslab_dx, int nb_slab, const float* __restrict__ dqp,
    const float* __restrict__ pbeta, const float* __restrict__ W_h /*[A][H]*/,
    const float* __restrict__ Gact, const float* __restrict__ Call, float* __restrict__ carry_dc,
    float* __restrict__ dG, float* __restrict__ dq_all, float* __restrict__ dinit) {
  __shared__ float dq_s[kA];
  float dh = 0.f, dc = 0.f;
  const bool carry = have_next && b < nb_next;          // row b was active at step t+1
  if (carry && active) {
    float q = 0.f;
    for (int c = 0; c < nlch; ++c) q += dqp[((long long)c * B + b) * kA + j];
    dq_s[j] = q;
    dq_all[((long long)b * T + (t + 1)) * kA + j] = q;
  }
  __syncthreads();
  if (!active) return;
  if (carry) {
    float s = 0.f;
#pragma unroll
    for (int z = 0; z < kS_DX; ++z) s += (z < nslab_dx) ? slab_dx[((long long)z * nb_slab + b) * kXK + kE + kD + j] : 0.f;
#pragma unroll
    for (int c = 0; c < kNCH; ++c) s += pbeta[((long long)c * B + b) * kH + j];
#pragma unroll 32
    for (int a = 0; a < kA; ++a) s += dq_s[a] * W_h[a * kH + j];
    dh = s;
    dc = carry_dc[b * kH + j];
  }
  if (final_pass) {
    dinit[b * 2 * kH + j] = dh;
    dinit[b * 2 * kH + kH + j] = dc;
    return;
  }
  const float dm = drop ? drop[((long long)b * T + t) * kH + j] : 1.0f;
  dh += dHd[((long long)packed_off + b) * kH + j] * dm;
  const float* ga = Gact + ((long long)b * T + t) * kG;
  const float ig = ga[j], fg = ga[kH + j], gg = ga[2 * kH + j], og = ga[3 * kH + j];
  const long long hc = ((long long)b * (T + 1) + t) * kH + j;
  const float cprev = Call[hc], tc = tanhf(Call[hc + kH]);
  const float dog = dh * tc;
  dc += dh * og * (1.f - tc * tc);
  carry_dc[b * kH + j] = dc * fg;
  float* dg = dG + ((long long)b * T + t) * kG;
  dg[j] = dc * gg * ig * (1.f - ig);
  dg[kH + j] = dc * cprev * fg * (1.f - fg);
  dg[2 * kH + j] = dc * ig * (1.f - gg * gg);
  dg[3 * kH + j] = dog * og * (1.f - og);
}

__global__ void __launch_bounds__(kH) lstm_bwd_kernel(
    int t, int T, int B, int nb_next, int have_next, int final_pass, int nlch,
    const float* __restrict__ dHd, int packed_off, const float* __restrict__ drop,
    const float* __restrict__ slab_dx, int nslab_dx, int nb_slab, const float* __restrict__ dqp,
    const float* __restrict__ pbeta, const float* __restrict__ W_h /*[A][H]*/,
    const float* __restrict__ Gact, const float* __restrict__ Call, float* __restrict__ carry_dc,
    float* __restrict__ dG, float* __restrict__ dq_all, float* __restrict__ dinit) {
  lstm_bwd_body(blockIdx.x, threadIdx.x, true, t, T, B, nb_next, have_next, final_pass, nlch, dHd, packed_off, drop, slab_dx,
                nslab_dx, nb_slab, dqp, pbeta, W_h, Gact, Call, carry_dc, dG, dq_all, dinit);
}

// ------------------------------------------------------------------------------------------
// backward step kernel 3 (grid kNCH x nb): gate / context gradients for one 256-channel chunk,
// the second pass over F[b,:,chunk] (d alpha partial), the W_beta^T dgpre partial for dh_{t-1},
// and (chunk 0) the embedding-row scatter.
// ------------------------------------------------------------------------------------------
template <int L>
__global__ void __launch_bounds__(512, 4) attn_bwd_a_kernel(
    const float* __restrict__ F, const float* __restrict__ slab_dx, int nslab, int nb, int B, int t, int T,
    const float* __restrict__ ctx_all, const float* __restrict__ gate_all, const float* __restrict__ W_beta,
    const long long* __restrict__ cap, int cap_stride, int V, float* __restrict__ dctx_all,
    float* __restrict__ dgpre_all, float* __restrict__ dalp, float* __restrict__ pbeta, float* __restrict__ dembed) {
  __shared__ __align__(16) float dctx_s[256];
  __shared__ float dgp_s[256];
  __shared__ float pb_s[4][kH];
  __shared__ float da_s[256];
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x;
  const long long bt = (long long)b * T + t;
  if (tid < 256) {
    const int d = chunk * 256 + tid;
    float dx = 0.f;
#pragma unroll
    for (int z = 0; z < kS_DX; ++z) dx += (z < nslab) ? slab_dx[((long long)z * nb + b) * kXK + kE + d] : 0.f;
    const float c = ctx_all[bt * kD + d], g = gate_all[bt * kD + d];
    const float dgp = dx * c * g * (1.f - g);
    const float dcx = dx * g;
    dgpre_all[bt * kD + d] = dgp;
    dctx_all[bt * kD + d] = dcx;
    dctx_s[tid] = dcx;
    dgp_s[tid] = dgp;
  } else if (chunk == 0 && tid < 256 + kE) {   // gradient of the embedded input row (b, t); summed per token after BPTT
    const int e = tid - 256;                   // by embed_grad_kernel in a fixed order (no atomics: bit-reproducible)
    float dx = 0.f;
#pragma unroll
    for (int z = 0; z < kS_DX; ++z) dx += (z < nslab) ? slab_dx[((long long)z * nb + b) * kXK + e] : 0.f;
    dembed[bt * kE + e] = dx;
  }
  __syncthreads();
  // Both remaining parts read long-latency data, so all their loads are issued before the first use:
  //  (1) partial of W_beta^T dgpre over this chunk's 256 channels: output k, four quarters of 64 channels;
  //  (2) d alpha partial: dot(dctx[chunk], F[b,l,chunk]); a 16-lane group per cell (lane covers channels
  //      ln*4 + 64*j, so each load instruction reads 256 contiguous bytes per cell), cells l = group + 32*i.
  const int wv_id = __builtin_amdgcn_readfirstlane(tid >> 6);        // scalar wave index -> uniform bases below
  const int lane = tid & 63;
  const int k = tid & (kH - 1), quarter = wv_id >> 1;
  const float* Wb = W_beta + ((long long)chunk * 256 + quarter * 64) * kH + (wv_id & 1) * 64;      // uniform
  const int ln = tid & 15, grp = tid >> 4;
  const float* Fu = F + (long long)b * L * kD + chunk * 256;                                       // uniform
  float4 dc4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) dc4[j] = *reinterpret_cast<const float4*>(&dctx_s[ln * 4 + 64 * j]);
  float ps = 0.f;
#pragma unroll 1
  for (int part = 0; part < 4; ++part) {          // 4 passes x (16 W_beta values + 2 cells x 64 B) per thread
    float wv[16];
    float4 v[2][4];
#pragma unroll
    for (int d = 0; d < 16; ++d) wv[d] = Wb[(part * 16 + d) * kH + lane];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int l = grp + 32 * (part * 2 + i);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        v[i][j] = (32 * (part * 2 + i) < L)      // (uniform: whole passes beyond the last cell are skipped)
                      ? *reinterpret_cast<const float4*>(Fu + (unsigned)min(l, L - 1) * kD + ln * 4 + 64 * j)   // branch-free guard
                      : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int d = 0; d < 16; ++d) ps += dgp_s[quarter * 64 + part * 16 + d] * wv[d];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float sacc = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        sacc += dc4[j].x * v[i][j].x + dc4[j].y * v[i][j].y + dc4[j].z * v[i][j].z + dc4[j].w * v[i][j].w;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o, 64);
      if (ln == 0) da_s[grp + 32 * (part * 2 + i)] = sacc;       // da_s is padded to 256 cells
    }
  }
  pb_s[quarter][k] = ps;
  __syncthreads();
  if (tid < kH)
    pbeta[((long long)chunk * B + b) * kH + tid] = (pb_s[0][tid] + pb_s[1][tid]) + (pb_s[2][tid] + pb_s[3][tid]);
  if (tid < L) dalp[((long long)chunk * B + b) * L + tid] = da_s[tid];
}

// ------------------------------------------------------------------------------------------
// d embed[token] = sum over the decoded rows (b, t) that fed this token, in increasing (b, t) order.  One workgroup
// (two waves, thread = embedding column) per row n: all rows are tested against n's token 128 at a time, the per-wave
// ballots go to LDS; the row that is the FIRST occurrence of its token then walks the set bits in order, adds those
// rows up and stores the result (the table was zeroed before); every other row exits.  No atomics: bit-reproducible.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kE) embed_grad_kernel(const float* __restrict__ dXe, const long long* __restrict__ cap,
                                                        int cap_stride, const int* __restrict__ dec_len, int B, int T,
                                                        int V, float* __restrict__ dembed) {
  extern __shared__ unsigned long long bal_s[];            // [chunks][2 waves]
  static_assert(kE == 128, "embed_grad_kernel: two waves of 64 columns");
  const int n = blockIdx.x, N = B * T, e = threadIdx.x, wave = e >> 6;
  const int bn = n / T, tn = n - bn * T;
  if (tn >= dec_len[bn]) return;                           // row not decoded (uniform)
  long long tokl = cap[(long long)bn * cap_stride + tn];
  const int tok = (int)(tokl < 0 ? 0 : (tokl >= V ? V - 1 : tokl));
  const int chunks = (N + kE - 1) / kE;
  for (int c0 = 0; c0 < chunks; c0 += 4) {                   // four chunks' token / length loads in flight together
    long long idv[4];
    int lenv[4], tv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int m = min((c0 + u) * kE + e, N - 1);           // clamped: branch-free loads, masked below
      const int b = m / T;
      tv[u] = m - b * T;
      idv[u] = cap[(long long)b * cap_stride + tv[u]];
      lenv[u] = dec_len[b];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + u;
      long long id = idv[u];
      id = id < 0 ? 0 : (id >= V ? V - 1 : id);
      const bool hit = c * kE + e < N && tv[u] < lenv[u] && (int)id == tok;
      const unsigned long long mask = __ballot(hit);
      if ((e & 63) == 0 && c < chunks) bal_s[c * 2 + wave] = mask;
    }
  }
  __syncthreads();
  float acc = 0.f;
  bool first = true;
  for (int w2 = 0; w2 < chunks * 2; ++w2) {                // masks in increasing row order
    unsigned long long mask = bal_s[w2];
    while (mask) {
      const int m = w2 * 64 + __builtin_ctzll(mask);
      if (first && m != n) return;                         // an earlier row carries this token: that row does the sum
      first = false;
      acc += dXe[(long long)m * kE + e];
      mask &= mask - 1;
    }
  }
  dembed[(long long)tok * kE + e] = acc;
}

// ------------------------------------------------------------------------------------------
// backward step kernel 4 (grid kLCH x nb): softmax / Gumbel-softmax backward, score backward over a
// 49-cell slice: dq partial, dP accumulation (P is time-invariant -> its gradient sums over steps),
// full_att weight/bias gradient accumulators (private per (slice,row): deterministic).
// ------------------------------------------------------------------------------------------
template <int L>
__device__ __forceinline__ void attn_bwd_b_body(
    const int lch, const int b,
    const float* __restrict__ P, const float* __restrict__ Qall, const float* __restrict__ alphas,
    const float* __restrict__ dalp, const float* __restrict__ dalphas_in, const float* __restrict__ w_full,
    int B, int t, int T, const int* __restrict__ dec_len, float inv_temp, float* __restrict__ dPacc,
    float* __restrict__ dqp, float* __restrict__ dwf_acc, float* __restrict__ dbf_acc) {
  __shared__ float de_s[L];
  __shared__ float red_s[4];
  __shared__ float dbf_s[8];
  __shared__ __align__(16) float acc_s[8][2][kA];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const long long bt = (long long)b * T + t;
  // BPTT runs t = T-1 .. 0; row b joins at its own last step, where its accumulators are initialised
  const bool first_step = (t == dec_len[b] - 1);
  float al = 0.f, da = 0.f;
  if (tid < L) {
    al = alphas[bt * L + tid];
#pragma unroll
    for (int c = 0; c < kNCH; ++c) da += dalp[((long long)c * B + b) * L + tid];
    if (dalphas_in) da += dalphas_in[bt * L + tid];
  }
  const float part = wave_sum(al * da);
  if (lane == 0) red_s[w] = part;
  __syncthreads();
  const float dot = red_s[0] + red_s[1] + red_s[2] + red_s[3];
  if (tid < L) de_s[tid] = al * (da - dot) * inv_temp;
  __syncthreads();
  const int l32 = lane & 31, sub = lane >> 5, hw = w * 2 + sub;      // 8 half-waves
  const float4 q4 = *reinterpret_cast<const float4*>(Qall + bt * kA + l32 * 4);
  const float4 w4 = *reinterpret_cast<const float4*>(w_full + l32 * 4);
  float4 dq4 = make_float4(0.f, 0.f, 0.f, 0.f), dw4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float dbf = 0.f;
  constexpr int SLICE = 49;                          // cells per workgroup (grid.x = L / 49 slices)
  const int l_lo = lch * SLICE, l_hi = l_lo + SLICE;
  constexpr int NIT = (SLICE + 7) / 8;          // 49 cells over 8 half-waves -> 7 passes, all loads up front
  float4 p4v[NIT], oldv[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int l = min(l_lo + hw + 8 * i, l_hi - 1);
    const long long o = ((long long)b * L + l) * kA + l32 * 4;
    p4v[i] = *reinterpret_cast<const float4*>(P + o);                 // clamped cell: branch-free, unused when l >= l_hi
    oldv[i] = *reinterpret_cast<const float4*>(dPacc + o);
    if (first_step) oldv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int l = l_lo + hw + 8 * i;
    if (l < l_hi) {
      const long long o = ((long long)b * L + l) * kA + l32 * 4;
      const float4 p4 = p4v[i];
      const float de = de_s[l];
      const float r0 = p4.x + q4.x, r1 = p4.y + q4.y, r2 = p4.z + q4.z, r3 = p4.w + q4.w;
      float4 dp;
      dp.x = r0 > 0.f ? de * w4.x : 0.f;
      dp.y = r1 > 0.f ? de * w4.y : 0.f;
      dp.z = r2 > 0.f ? de * w4.z : 0.f;
      dp.w = r3 > 0.f ? de * w4.w : 0.f;
      dq4.x += dp.x; dq4.y += dp.y; dq4.z += dp.z; dq4.w += dp.w;
      dw4.x += de * fmaxf(r0, 0.f); dw4.y += de * fmaxf(r1, 0.f);
      dw4.z += de * fmaxf(r2, 0.f); dw4.w += de * fmaxf(r3, 0.f);
      if (l32 == 0) dbf += de;
      float4 acc = dp;
      acc.x += oldv[i].x; acc.y += oldv[i].y; acc.z += oldv[i].z; acc.w += oldv[i].w;
      *reinterpret_cast<float4*>(dPacc + o) = acc;
    }
  }
  *reinterpret_cast<float4*>(&acc_s[hw][0][l32 * 4]) = dq4;
  *reinterpret_cast<float4*>(&acc_s[hw][1][l32 * 4]) = dw4;
  if (l32 == 0) dbf_s[hw] = dbf;
  __syncthreads();
  if (tid < kA) {
    float s = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s += acc_s[i][0][tid]; s2 += acc_s[i][1][tid]; }
    const long long o = ((long long)lch * B + b) * kA + tid;
    dqp[o] = s;
    dwf_acc[o] = (first_step ? 0.f : dwf_acc[o]) + s2;
  }
  if (tid == 0) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += dbf_s[i];
    const long long o = (long long)lch * B + b;
    dbf_acc[o] = (first_step ? 0.f : dbf_acc[o]) + s;
  }
}

template <int L>
__global__ void __launch_bounds__(256) attn_bwd_b_kernel(
    const float* __restrict__ P, const float* __restrict__ Qall, const float* __restrict__ alphas,
    const float* __restrict__ dalp, const float* __restrict__ dalphas_in, const float* __restrict__ w_full,
    int B, int t, int T, const int* __restrict__ dec_len, float inv_temp, float* __restrict__ dPacc,
    float* __restrict__ dqp, float* __restrict__ dwf_acc, float* __restrict__ dbf_acc) {
  attn_bwd_b_body<L>(blockIdx.x, blockIdx.y, P, Qall, alphas, dalp, dalphas_in, w_full, B, t, T, dec_len, inv_temp, dPacc, dqp,
                     dwf_acc, dbf_acc);
}

// Compact layout (one score slice per row): the score backward of step t and the LSTM-cell backward of step t-1 (which
// consumes its dq) in one launch, one workgroup per row that is active at step t-1 (or every row for the closing
// h0/c0 pass); rows that ended before step t skip the first half.  One dependent launch less per BPTT step.
struct LstmBwdArgs {
  int t, T, B, nb_next, have_next, final_pass, nlch;
  const float* dHd; int packed_off; const float* drop;
  const float* slab_dx; int nslab_dx, nb_slab; const float* dqp;
  const float* pbeta; const float* W_h;
  const float* Gact; const float* Call; float* carry_dc;
  float* dG; float* dq_all; float* dinit;
};
template <int L>
__global__ void __launch_bounds__(256) attn_bwd_b_lstm_kernel(
    const float* __restrict__ P, const float* __restrict__ Qall, const float* __restrict__ alphas,
    const float* __restrict__ dalp, const float* __restrict__ dalphas_in, const float* __restrict__ w_full,
    int B, int t, int T, const int* __restrict__ dec_len, float inv_temp, float* __restrict__ dPacc,
    float* __restrict__ dqp, float* __restrict__ dwf_acc, float* __restrict__ dbf_acc, int nb_t, const LstmBwdArgs la) {
  const int b = blockIdx.x;
  if (b < nb_t)
    attn_bwd_b_body<L>(0, b, P, Qall, alphas, dalp, dalphas_in, w_full, B, t, T, dec_len, inv_temp, dPacc, dqp, dwf_acc, dbf_acc);
  __threadfence_block();             // this row's dq partial (global) is read back by the LSTM half below
  __syncthreads();
  lstm_bwd_body(b, threadIdx.x, threadIdx.x < kH, la.t, la.T, la.B, la.nb_next, la.have_next, la.final_pass, la.nlch, la.dHd,
                la.packed_off, la.drop, la.slab_dx, la.nslab_dx, la.nb_slab, la.dqp, la.pbeta, la.W_h, la.Gact, la.Call,
                la.carry_dc, la.dG, la.dq_all, la.dinit);
}

// ------------------------------------------------------------------------------------------
// dF[b,l,d] = sum_t alpha[b,t,l] * dctx[b,t,d] + dmean[b,d] / L     (the W_z^T dP term is added by an
// accumulating MFMA GEMM afterwards).  grid (kNCH, B); dctx of this thread's channel in registers.
// ------------------------------------------------------------------------------------------
template <int L, int TMAXR>
__global__ void __launch_bounds__(256) dF_init_kernel(const float* __restrict__ alphas, const float* __restrict__ dctx_all,
                                                       const float* __restrict__ dmean, int T, int Tb_unused,
                                                       const int* __restrict__ dec_len, float* __restrict__ dF) {
  extern __shared__ __align__(16) float al_s[];   // [L][TMAXR]
  const int chunk = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int d = chunk * 256 + tid;
  const int Tb = min(dec_len[b], T);
  for (int i = tid; i < L * TMAXR; i += 256) {
    const int l = i / TMAXR, tt = i - l * TMAXR;
    al_s[i] = (tt < Tb) ? alphas[((long long)b * T + tt) * L + l] : 0.f;
  }
  float dc[TMAXR];
#pragma unroll
  for (int tt = 0; tt < TMAXR; ++tt) dc[tt] = (tt < Tb) ? dctx_all[((long long)b * T + tt) * kD + d] : 0.f;
  const float dm = dmean[(long long)b * kD + d] / (float)L;
  __syncthreads();
  float* o = dF + (long long)b * L * kD + d;
  for (int l = 0; l < L; ++l) {
    float s = dm;
#pragma unroll
    for (int t4 = 0; t4 < TMAXR; t4 += 4) {
      const float4 a = *reinterpret_cast<const float4*>(&al_s[l * TMAXR + t4]);
      s = fmaf(a.x, dc[t4], s); s = fmaf(a.y, dc[t4 + 1], s); s = fmaf(a.z, dc[t4 + 2], s); s = fmaf(a.w, dc[t4 + 3], s);
    }
    o[(long long)l * kD] = s;
  }
}

// ------------------------------------------------------------------------------------------
// Backward of the stand-alone attention module (autograd of Soft_Attention.forward / Hard_Attention.forward,
// attention.py:81-95, 132-148): one workgroup per batch row.  Not on the training hot path (the decoders fuse their
// attention into the step kernels); written for clarity, every reduction in a fixed order.
//   d alpha_l  = dalpha_l + F_l . dctx              ctx = sum_l alpha_l F_l
//   d e_l      = alpha_l (d alpha_l - sum_j alpha_j d alpha_j) / temp
//   d pre[l,a] = d e_l w[a] [P[l,a] + q[a] > 0]     e_l = w . relu(P_l + q) + b,  q = W_h h + b_h
//   outputs: dP [L,A] (-> dW_z, db_z, dF += dP W_z by GEMMs), dq [A], per-row partials of dw / db, dF_l = alpha_l dctx
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) attention_bwd_kernel(
    const float* __restrict__ F, const float* __restrict__ P, const float* __restrict__ h, const float* __restrict__ W_h,
    const float* __restrict__ b_h, const float* __restrict__ w_full, const float* __restrict__ alpha,
    const float* __restrict__ dctx, const float* __restrict__ dalpha, float inv_temp, float* __restrict__ dP,
    float* __restrict__ dq, float* __restrict__ dwf_part, float* __restrict__ dbf_part, float* __restrict__ dF) {
  __shared__ __align__(16) float dctx_s[kD];
  __shared__ float q_s[kA], h_s[kH], da_s[kL], de_s[kL], red_s[4];
  __shared__ float acc_s[2][2][kA];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int d = tid; d < kD; d += 256) dctx_s[d] = dctx[(long long)b * kD + d];
  if (tid < kH) h_s[tid] = h[(long long)b * kH + tid];
  __syncthreads();
  if (tid < kA) {
    float q = b_h[tid];
    for (int k = 0; k < kH; ++k) q += W_h[tid * kH + k] * h_s[k];
    q_s[tid] = q;
  }
  const float* Fb = F + (long long)b * kL * kD;
  for (int l = w; l < kL; l += 4) {            // d alpha: one wave per cell, lanes stride the 2048 channels
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < kD / 256; ++j) {
      const float4 f = *reinterpret_cast<const float4*>(Fb + (long long)l * kD + j * 256 + lane * 4);
      const float4 g = *reinterpret_cast<const float4*>(&dctx_s[j * 256 + lane * 4]);
      s += f.x * g.x + f.y * g.y + f.z * g.z + f.w * g.w;
    }
    s = wave_sum(s);
    if (lane == 0) da_s[l] = s + (dalpha ? dalpha[(long long)b * kL + l] : 0.f);
  }
  __syncthreads();
  float al = 0.f, da = 0.f;
  if (tid < kL) { al = alpha[(long long)b * kL + tid]; da = da_s[tid]; }
  const float part = wave_sum(al * da);
  if (lane == 0) red_s[w] = part;
  __syncthreads();
  const float dot = (red_s[0] + red_s[1]) + (red_s[2] + red_s[3]);
  if (tid < kL) de_s[tid] = al * (da - dot) * inv_temp;
  __syncthreads();
  {  // score backward: thread (half, a) walks half of the cells
    const int a = tid & (kA - 1), half = tid >> 7;
    const float qa = q_s[a], wa = w_full[a];
    float sq = 0.f, sw = 0.f;
    for (int l = half * (kL / 2); l < (half + 1) * (kL / 2); ++l) {
      const long long o = ((long long)b * kL + l) * kA + a;
      const float r = P[o] + qa, de = de_s[l];
      const float dp = r > 0.f ? de * wa : 0.f;
      dP[o] = dp;
      sq += dp;
      sw += de * fmaxf(r, 0.f);
    }
    acc_s[half][0][a] = sq;
    acc_s[half][1][a] = sw;
  }
  __syncthreads();
  if (tid < kA) {
    dq[(long long)b * kA + tid] = acc_s[0][0][tid] + acc_s[1][0][tid];
    dwf_part[(long long)b * kA + tid] = acc_s[0][1][tid] + acc_s[1][1][tid];
  }
  if (tid == 0) {
    float s = 0.f;
    for (int l = 0; l < kL; ++l) s += de_s[l];
    dbf_part[b] = s;
  }
  float* dFb = dF + (long long)b * kL * kD;     // dF_l = alpha_l * dctx   (the W_z^T dP term is accumulated by a GEMM)
  for (int l = 0; l < kL; ++l) {
    const float a_l = alpha[(long long)b * kL + l];
#pragma unroll
    for (int j = 0; j < kD / 1024; ++j) {
      const float4 g = *reinterpret_cast<const float4*>(&dctx_s[j * 1024 + tid * 4]);
      *reinterpret_cast<float4*>(dFb + (long long)l * kD + j * 1024 + tid * 4) = make_float4(a_l * g.x, a_l * g.y, a_l * g.z, a_l * g.w);
    }
  }
}

// ------------------------------------------------------------------------------------------
// greedy decoding helpers (batch_sample / sample, depth_models.py:216-305): everything stays on the device
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fill_ids_kernel(long long* ids, int n, long long v) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) ids[i] = v;
}

__global__ void __launch_bounds__(kE) embed_step_kernel(const float* __restrict__ embed, const long long* __restrict__ ids,
                                                         int t, int T, int V, float* __restrict__ Xall) {
  const int b = blockIdx.x;
  long long id = ids[b];
  id = id < 0 ? 0 : (id >= V ? V - 1 : id);
  Xall[((long long)b * T + t) * kXK + threadIdx.x] = embed[id * kE + threadIdx.x];
}

// ids[b] = argmax_v logits[b,v] (first maximum on ties, like torch.argmax); also out[b*T + t]
__global__ void __launch_bounds__(256) argmax_kernel(const float* __restrict__ logits, int V, int t, int T,
                                                      long long* __restrict__ ids, long long* __restrict__ out) {
  __shared__ float bv[4];
  __shared__ int bi[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* x = logits + (long long)b * V;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  for (int v = tid; v < V; v += 256) {
    const float f = x[v];
    if (f > best) { best = f; idx = v; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (ob > best || (ob == best && oi < idx)) { best = ob; idx = oi; }
  }
  if (lane == 0) { bv[w] = best; bi[w] = idx; }
  __syncthreads();
  if (tid == 0) {
    for (int i = 1; i < 4; ++i)
      if (bv[i] > best || (bv[i] == best && bi[i] < idx)) { best = bv[i]; idx = bi[i]; }
    ids[b] = idx;
    out[(long long)b * T + t] = idx;
  }
}

// ------------------------------------------------------------------------------------------
// host orchestration
// ------------------------------------------------------------------------------------------
struct StepPlan {
  int T = 0, N = 0;
  std::vector<int> bs, off;
};

static int make_plan(const int* dec_len, int B, StepPlan* pl) {
  DIC_REQUIRE(B > 0 && dec_len != nullptr, "decoder: empty batch");
  for (int b = 1; b < B; ++b)
    DIC_REQUIRE(dec_len[b] <= dec_len[b - 1], "decoder: lengths must be sorted in descending order (util.py:95)");
  DIC_REQUIRE(dec_len[B - 1] >= 1, "decoder: every caption needs at least one decode step");
  pl->T = dec_len[0];
  pl->bs.assign(pl->T, 0);
  pl->off.assign(pl->T + 1, 0);
  for (int t = 0; t < pl->T; ++t) {
    int nb = 0;
    for (int b = 0; b < B; ++b) nb += dec_len[b] > t;
    pl->bs[t] = nb;
    pl->off[t + 1] = pl->off[t] + nb;
  }
  pl->N = pl->off[pl->T];
  return DIC_OK;
}

static int launch_transpose(const float* in, float* out, int R, int Cc, hipStream_t st) {
  hipLaunchKernelGGL(transpose_kernel, dim3(ceil_div(Cc, 32), ceil_div(R, 32)), dim3(256), 0, st, in, out, R, Cc);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

static int gemm(int M, int N, int K, GemmOperand A, GemmOperand B, GemmEpilogue ep, hipStream_t st, int splitk = 1,
                float* ws = nullptr, int tile = 0) {
  GemmParams p{};
  p.M = M; p.N = N; p.K = K; p.A = A; p.B = B; p.ep = ep; p.splitk = splitk; p.ws = ws;
  return gemm_launch(p, st, tile);
}

// ------------------------------------------------------------------------------------------
// Compact (49-cell) mode.  At 224x224 both encoders end in a 7x7 map that AdaptiveAvgPool2d(14) replicates 2x2 exactly
// (quirk Q3), so the 196 annotation cells hold 49 distinct vectors.  Equal scores within a group make
// softmax_196 = softmax_49 / 4 and ctx = sum_g beta_g F_g: the decoder runs on the 49 distinct cells (every pass over
// F and P is 4x smaller) and only the returned alphas are expanded / the incoming alpha gradient is folded.
// Group g = (i, j) of the 7x7 map <-> cells (2i + di) * 14 + (2j + dj).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) expand_alphas_kernel(const float* __restrict__ ac, float* __restrict__ a,
                                                             long long n) {       // n = B*T*196
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long long bt = i / kL;
  const int cell = (int)(i - bt * kL);
  const int g = (cell / 28) * 7 + (cell % 14) / 2;
  a[i] = 0.25f * ac[bt * kLc + g];
}
__global__ void __launch_bounds__(256) fold_dalphas_kernel(const float* __restrict__ da, float* __restrict__ dc,
                                                            long long n) {        // n = B*T*49
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long long bt = i / kLc;
  const int g = (int)(i - bt * kLc);
  const float* r = da + bt * kL + (g / 7) * 28 + (g % 7) * 2;
  dc[i] = 0.25f * ((r[0] + r[1]) + (r[14] + r[15]));
}

// The persistent forward loop (csrc/experiments/decoder_persist.hip, switch 141) is a parked experiment: correct, but at
// batch 64 it takes 20 us per step against 21.7 us for the two launches it replaces.  It exists only in the
// experiments build; the product library always runs the per-step launches.
#ifdef DIC_EXPERIMENTS
static int g_persistent = 0;
void decoder_debug_persistent(int on) { g_persistent = on; }
#endif

// runs STMT with a compile-time cell count L_ (196 = reference layout, 49 = compact)
#define DIC_CELLS_SWITCH(CELLS, STMT)   \
  if ((CELLS) == kL) {                  \
    constexpr int L_ = kL;              \
    STMT                                \
  } else {                              \
    constexpr int L_ = kLc;             \
    STMT                                \
  }

// raw split-K partial slabs [splitk][M][N] (no reduce launch; the consumer kernel sums the slabs)
static int gemm_slabs(int M, int N, int K, GemmOperand A, GemmOperand B, float* slabs, int splitk, hipStream_t st) {
  GemmParams p{};
  p.M = M; p.N = N; p.K = K; p.A = A; p.B = B; p.ep = ep_store(slabs, N); p.splitk = splitk; p.ws = slabs;
  p.raw_partials = 1;
  return gemm_launch(p, st, 64);
}

}  // namespace dic

using namespace dic;

static int check_common(const dic_decoder_weights* w, int V, int B, const void* ws) {
  DIC_REQUIRE(w != nullptr && ws != nullptr, "decoder: null weights/workspace");
  DIC_REQUIRE(V > 0 && B > 0, "decoder: bad sizes");
  return DIC_OK;
}

extern "C" {

int dic_decoder_inspect(const void* workspace, size_t workspace_bytes, int B, int Tmax, int V, int n_packed, int cells, int which,
                        float* out, long long* n_out, void* stream) {
  DIC_REQUIRE(workspace && B > 0 && Tmax > 0 && (which == 1 || which == 2) && (cells == kL || cells == kLc), "decoder_inspect: bad arguments");
  bool ov = false;
  DecoderWs ws = decoder_carve(const_cast<void*>(workspace), workspace_bytes, B, Tmax, V, n_packed, &ov);
  DIC_REQUIRE(!ov, "decoder_inspect: workspace too small");
  const long long n = which == 1 ? (long long)B * cells * kA : (long long)B * Tmax * kA;
  if (n_out) *n_out = n;
  if (!out) return DIC_OK;
  DIC_CHECK_HIP(hipMemcpyAsync(out, which == 1 ? ws.P : ws.Qall, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return DIC_OK;
}

size_t dic_decoder_workspace_bytes(int B, int Tmax, int V, int n_packed) {
  bool ov;
  return decoder_carve(nullptr, 0, B, Tmax, V, n_packed, &ov).bytes;
}

static int decoder_fwd_impl(const dic_decoder_weights* w, int V, const float* feat_rgb, const float* feat_depth,
                            const int64_t* captions, int cap_stride, const int* dec_lengths, int B,
                            const float* drop_mult, int mode, const float* gumbel_u, float temp, float* logits_packed,
                            float* alphas_out, void* workspace, size_t workspace_bytes, void* stream, int cells) {
  hipStream_t st = (hipStream_t)stream;
  DIC_REQUIRE(cells == kL || cells == kLc, "decoder_fwd: cells must be 196 or 49");
  DIC_REQUIRE(cells == kL || mode == 0, "decoder_fwd: the compact 49-cell layout needs soft attention (per-cell Gumbel "
                                        "noise breaks the 2x2 symmetry)");
  float* const alphas_out_ = alphas_out;
  DIC_TRY(check_common(w, V, B, workspace));
  DIC_REQUIRE(feat_rgb && captions && logits_packed && alphas_out, "decoder_fwd: null pointer");
  DIC_REQUIRE(mode >= 0 && mode <= 2, "decoder_fwd: mode must be 0 (soft), 1 (gumbel-softmax) or 2 (gumbel-max)");
  DIC_REQUIRE(mode == 0 || gumbel_u != nullptr, "decoder_fwd: hard attention needs the uniform draws");
  StepPlan pl;
  DIC_TRY(make_plan(dec_lengths, B, &pl));
  const int T = pl.T, N = pl.N;
  bool ov = false;
  DecoderWs ws = decoder_carve(workspace, workspace_bytes, B, T, V, N, &ov);
  DIC_REQUIRE(!ov, "decoder_fwd: workspace too small (%zu < %zu)", workspace_bytes, ws.bytes);

  int* d_len = ws.dlen;                                  // device copy of dec_lengths
  DIC_CHECK_HIP(hipMemcpyAsync(d_len, dec_lengths, sizeof(int) * B, hipMemcpyHostToDevice, st));
  float* alphas = (cells == kL) ? alphas_out_ : ws.alpha_c;      // [B,T,cells]: what the step kernels write
  DIC_CHECK_HIP(hipMemsetAsync(alphas, 0, sizeof(float) * (size_t)B * T * cells, st));
  DIC_CHECK_HIP(hipMemsetAsync(ws.Xall, 0, sizeof(float) * (size_t)B * T * kXK, st));

  // weight prep: fused LSTM weight, transposed small matrices for coalesced mat-vecs
  hipLaunchKernelGGL(pack_lstm_kernel, dim3(kG), dim3(256), 0, st, w->w_ih, w->w_hh, w->b_ih, w->b_hh, ws.Wcat, ws.bcat);
  DIC_TRY(launch_transpose(ws.Wcat, ws.WcatT, kG, kXK, st));      // [kXK][4H]: K-contiguous rows for the backward dX GEMM
  DIC_TRY(launch_transpose(w->dec_att_w, ws.WhT, kA, kH, st));
  DIC_TRY(launch_transpose(w->fbeta_w, ws.WbT, kD, kH, st));
  // F = F_rgb + F_depth, mean over cells
  DIC_CELLS_SWITCH(cells, hipLaunchKernelGGL(fuse_mean_kernel<L_>, dim3(kNCH, B), dim3(256), 0, st, feat_rgb, feat_depth,
                                             ws.F, ws.mean);)
  DIC_LAUNCH_CHECK();
  // P = Wz F + bz  (hoisted: time-invariant, quirk Q4)
  //   (compact layout: only 98 output tiles at batch 64 -> split K four ways to fill the chip)
  const int psplit = (cells == kL || (size_t)4 * B * cells * kA > ws.gemm_ws_floats) ? 1 : 4;
  DIC_TRY(gemm(B * cells, kA, kD, op_rowk(ws.F, kD), op_rowk(w->enc_att_w, kD), ep_store(ws.P, kA, w->enc_att_b), st, psplit,
               ws.gemm_ws));
  // [h0 | c0] = init_linear(mean)  -> slot 0 of Hall / Call
  {
    GemmEpilogue ep = ep_store(ws.Hall, (long long)(T + 1) * kH, w->init_b);
    ep.C2 = ws.Call; ep.ldc2 = (long long)(T + 1) * kH; ep.nsplit = kH;
    DIC_TRY(gemm(B, 2 * kH, kD, op_rowk(ws.mean, kD), op_rowk(w->init_w, kD), ep, st, 16, ws.gemm_ws, 64));
  }
  hipLaunchKernelGGL(embed_gather_kernel, dim3(T, B), dim3(kE), 0, st, w->embed, (const long long*)captions, cap_stride,
                     d_len, T, V, ws.Xall);
  DIC_LAUNCH_CHECK();

#ifdef DIC_EXPERIMENTS
  const bool persistent = g_persistent && decoder_persist_eligible(B, T, mode);
  if (persistent) {
    // embedding part of every step's gate pre-activations (time-invariant under teacher forcing) + (b_ih + b_hh)
    DIC_TRY(gemm(B * T, kG, kE, op_rowk(ws.Xall, kXK), op_rowk(ws.Wcat, kXK), ep_store(ws.Gemb, kG, ws.bcat), st));
    DIC_TRY(decoder_fwd_persistent(ws, w, B, T, cells, drop_mult, alphas, pl.off.data(), st));
  }
#else
  constexpr bool persistent = false;
#endif
  for (int t = 0; t < T && !persistent; ++t) {
    const int nb = pl.bs[t];
    // steps t >= 1 carry the LSTM cell of step t-1 in their prologue (FusedLstm); rows that ended at t-1 are
    // still in the grid (nb_prev >= nb) for that part only
    FusedLstm fl{};
    int rows = nb;
    if (t > 0) {
      fl = FusedLstm{ws.slab_g, ws.bcat, drop_mult, ws.Hall, ws.Call, ws.Gact, ws.Hdrop, kS_LSTM, pl.bs[t - 1], nb,
                     pl.off[t - 1]};
      rows = pl.bs[t - 1];
    }
    DIC_CELLS_SWITCH(cells, hipLaunchKernelGGL(attn_fwd_kernel<L_>, dim3(kNCH, (rows + 7) / 8 * 8), dim3(512), 0, st, ws.F, ws.P, ws.Hall,
                                               ws.WhT, w->dec_att_b, w->full_att_w, w->full_att_b, ws.WbT, w->fbeta_b, t,
                                               T, mode, gumbel_u, B, temp, alphas, ws.Qall, ws.ctx, ws.gate, ws.Xall, 1,
                                               fl, rows);)
    DIC_LAUNCH_CHECK();
    DIC_TRY(gemm_slabs(nb, kG, kXK, op_rowk(ws.Xall + (long long)t * kXK, (long long)T * kXK), op_rowk(ws.Wcat, kXK),
                       ws.slab_g, kS_LSTM, st));
  }
  if (T > 0 && !persistent) {       // the last step's cell has no following attention launch
    const int tl = T - 1;
    hipLaunchKernelGGL(lstm_fwd_kernel, dim3(pl.bs[tl]), dim3(kH), 0, st, ws.slab_g, kS_LSTM, pl.bs[tl], ws.bcat, tl, T,
                       drop_mult, pl.off[tl], ws.Hall, ws.Call, ws.Gact, ws.Hdrop);
    DIC_LAUNCH_CHECK();
  }
  if (cells != kL) {      // returned attention weights in the reference's 196-cell layout: alpha_cell = beta_group / 4
    const long long n = (long long)B * T * kL;
    hipLaunchKernelGGL(expand_alphas_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ws.alpha_c, alphas_out_, n);
    DIC_LAUNCH_CHECK();
  }
  // logits (time-major packed rows) = dropout(h) W_o^T + b_o    (depth_models.py:197,204)
  DIC_TRY(gemm(N, V, kH, op_rowk(ws.Hdrop, kH), op_rowk(w->out_w, kH), ep_store(logits_packed, V, w->out_b), st));
  return DIC_OK;
}

extern "C" int dic_decoder_fwd(const dic_decoder_weights* w, int V, const float* feat_rgb, const float* feat_depth,
                               const int64_t* captions, int cap_stride, const int* dec_lengths, int B,
                               const float* drop_mult, int mode, const float* gumbel_u, float temp, float* logits_packed,
                               float* alphas, void* workspace, size_t workspace_bytes, void* stream) {
  return decoder_fwd_impl(w, V, feat_rgb, feat_depth, captions, cap_stride, dec_lengths, B, drop_mult, mode, gumbel_u, temp,
                          logits_packed, alphas, workspace, workspace_bytes, stream, kL);
}

extern "C" int dic_decoder_fwd_cells(const dic_decoder_weights* w, int V, const float* feat_rgb, const float* feat_depth,
                                     int cells, const int64_t* captions, int cap_stride, const int* dec_lengths, int B,
                                     const float* drop_mult, float* logits_packed, float* alphas, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  return decoder_fwd_impl(w, V, feat_rgb, feat_depth, captions, cap_stride, dec_lengths, B, drop_mult, 0, nullptr, 1.0f,
                          logits_packed, alphas, workspace, workspace_bytes, stream, cells);
}

static int decoder_bwd_impl(const dic_decoder_weights* w, int V, const int64_t* captions, int cap_stride,
                            const int* dec_lengths, int B, const float* drop_mult, int mode, float temp,
                            const float* dlogits_packed, const float* dalphas_in, const float* alphas_in,
                            const dic_decoder_grads* g, float* d_features, void* workspace, size_t workspace_bytes,
                            void* stream, int cells) {
  hipStream_t st = (hipStream_t)stream;
  DIC_REQUIRE(cells == kL || (cells == kLc && mode == 0), "decoder_bwd: cells must be 196, or 49 with soft attention");
  const float* alphas = alphas_in;
  const float* dalphas = dalphas_in;
  const int nlch = cells / 49;                 // score-backward slices of 49 cells
  DIC_TRY(check_common(w, V, B, workspace));
  DIC_REQUIRE(g && dlogits_packed && alphas_in && captions, "decoder_bwd: null pointer");
  DIC_REQUIRE(mode == 0 || mode == 1, "decoder_bwd: only soft (0) and gumbel-softmax (1) attention are differentiable");
  StepPlan pl;
  DIC_TRY(make_plan(dec_lengths, B, &pl));
  const int T = pl.T, N = pl.N;
  DIC_REQUIRE(T <= 64, "decoder_bwd: at most 64 decode steps supported (got %d)", T);
  bool ov = false;
  DecoderWs ws = decoder_carve(workspace, workspace_bytes, B, T, V, N, &ov);
  DIC_REQUIRE(!ov, "decoder_bwd: workspace too small");
  const size_t BT = (size_t)B * T;

  int* d_len = ws.dlen;
  DIC_CHECK_HIP(hipMemcpyAsync(d_len, dec_lengths, sizeof(int) * B, hipMemcpyHostToDevice, st));
  // rows that ended early keep zero gradients: dG, dctx, dgpre, dq are carved back to back (decoder_carve)
  DIC_CHECK_HIP(hipMemsetAsync(ws.dG, 0, (size_t)((char*)(ws.dq + BT * kA) - (char*)ws.dG), st));
  DIC_CHECK_HIP(hipMemsetAsync(g->embed, 0, sizeof(float) * (size_t)V * kE, st));
  float* cs = ws.colsum_ws;
  if (cells != kL) {      // compact layout: beta = group softmax saved by the forward; fold the 196-cell alpha gradient
    alphas = ws.alpha_c;
    if (dalphas_in) {
      const long long n = (long long)B * T * kLc;
      hipLaunchKernelGGL(fold_dalphas_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dalphas_in, ws.dalpha_c, n);
      DIC_LAUNCH_CHECK();
      dalphas = ws.dalpha_c;
    }
  }

  // vocabulary projection backward (batched over all steps)
  DIC_TRY(gemm(N, kH, V, op_rowk(dlogits_packed, V), op_colk(w->out_w, kH), ep_store(ws.dHd, kH), st, 8, ws.gemm_ws));
  DIC_TRY(gemm(V, kH, N, op_colk(dlogits_packed, V), op_colk(ws.Hdrop, kH), ep_store(g->out_w, kH), st));
  DIC_TRY(colsum(dlogits_packed, V, N, V, g->out_b, cs, st));

  const float inv_temp = (mode == 1) ? 1.0f / temp : 1.0f;
  // BPTT.  Per step: LSTM-cell backward (t) -> dX GEMM -> attention backward a -> attention backward b.  In the compact
  // layout the b half of step t shares its launch with the LSTM-cell backward of step t-1 (attn_bwd_b_lstm_kernel).
  const bool fuse_b = (nlch == 1);
  auto lstm_args = [&](int t) {       // arguments of the LSTM-cell backward of step t (t = -1: closing h0/c0 pass)
    LstmBwdArgs a{};
    a.T = T; a.B = B; a.nlch = nlch; a.dHd = ws.dHd; a.drop = drop_mult; a.slab_dx = ws.slab_dx; a.nslab_dx = kS_DX;
    a.dqp = ws.dqp; a.pbeta = ws.pbeta; a.W_h = w->dec_att_w; a.Gact = ws.Gact; a.Call = ws.Call; a.carry_dc = ws.carry_dc;
    a.dG = ws.dG; a.dq_all = ws.dq; a.dinit = ws.dinit;
    if (t < 0) { a.t = -1; a.nb_next = pl.bs[0]; a.have_next = 1; a.final_pass = 1; a.packed_off = 0; a.nb_slab = pl.bs[0]; }
    else {
      a.t = t; a.have_next = (t + 1 < T); a.nb_next = a.have_next ? pl.bs[t + 1] : 0; a.final_pass = 0;
      a.packed_off = pl.off[t]; a.nb_slab = a.nb_next;
    }
    return a;
  };
  auto launch_lstm = [&](int t) {
    const LstmBwdArgs a = lstm_args(t);
    hipLaunchKernelGGL(lstm_bwd_kernel, dim3(t < 0 ? B : pl.bs[t]), dim3(kH), 0, st, a.t, a.T, a.B, a.nb_next, a.have_next,
                       a.final_pass, a.nlch, a.dHd, a.packed_off, a.drop, a.slab_dx, a.nslab_dx, a.nb_slab, a.dqp, a.pbeta,
                       a.W_h, a.Gact, a.Call, a.carry_dc, a.dG, a.dq_all, a.dinit);
  };
  launch_lstm(T - 1);
  DIC_LAUNCH_CHECK();
  for (int t = T - 1; t >= 0; --t) {
    const int nb = pl.bs[t];
    // dX = dG_t * Wcat  (K = 4H)
    DIC_TRY(gemm_slabs(nb, kXK, kG, op_rowk(ws.dG + (long long)t * kG, (long long)T * kG), op_rowk(ws.WcatT, kG),
                       ws.slab_dx, kS_DX, st));
    DIC_CELLS_SWITCH(cells, hipLaunchKernelGGL(attn_bwd_a_kernel<L_>, dim3(kNCH, nb), dim3(512), 0, st, ws.F, ws.slab_dx,
                                               kS_DX, nb, B, t, T, ws.ctx, ws.gate, w->fbeta_w,
                                               (const long long*)captions, cap_stride, V, ws.dctx, ws.dgpre, ws.dalp,
                                               ws.pbeta, ws.dXe);)
    DIC_LAUNCH_CHECK();
    if (fuse_b) {     // score backward of step t + LSTM-cell backward of step t-1 (t = 0: the closing h0/c0 pass)
      const LstmBwdArgs la = lstm_args(t - 1);
      const int rows = t > 0 ? pl.bs[t - 1] : B;
      DIC_CELLS_SWITCH(cells, hipLaunchKernelGGL(attn_bwd_b_lstm_kernel<L_>, dim3(rows), dim3(256), 0, st, ws.P, ws.Qall,
                                                 alphas, ws.dalp, dalphas, w->full_att_w, B, t, T, ws.dlen, inv_temp,
                                                 ws.dPacc, ws.dqp, ws.dwf_acc, ws.dbf_acc, nb, la);)
      DIC_LAUNCH_CHECK();
    } else {
      DIC_CELLS_SWITCH(cells, hipLaunchKernelGGL(attn_bwd_b_kernel<L_>, dim3(nlch, nb), dim3(256), 0, st, ws.P, ws.Qall, alphas,
                                                 ws.dalp, dalphas, w->full_att_w, B, t, T, ws.dlen, inv_temp, ws.dPacc,
                                                 ws.dqp, ws.dwf_acc, ws.dbf_acc);)
      DIC_LAUNCH_CHECK();
      launch_lstm(t - 1);               // step t-1, or the gradient of (h0 | c0) and the dq of step 0 after t = 0
      DIC_LAUNCH_CHECK();
    }
  }
  // embedding gradient: per-token sum of the per-row gradients in a fixed order
  const size_t bal_bytes = (size_t)((B * T + kE - 1) / kE) * 2 * sizeof(unsigned long long);
  DIC_REQUIRE(bal_bytes <= 60 * 1024, "decoder_bwd: B*T too large for the embedding-gradient kernel");
  hipLaunchKernelGGL(embed_grad_kernel, dim3(B * T), dim3(kE), bal_bytes, st, ws.dXe,
                     (const long long*)captions, cap_stride, d_len, B, T, V, g->embed);
  DIC_LAUNCH_CHECK();

  // ---- bias gradients: seven column sums in two launches ------------------------------------------
  {
    ColsumBatch cb{};
    cb.j[0] = ColsumJob{ws.dG, kG, (int)BT, kG, 0, g->b_ih, nullptr};
    cb.j[1] = ColsumJob{ws.dgpre, kD, (int)BT, kD, 0, g->fbeta_b, nullptr};
    cb.j[2] = ColsumJob{ws.dq, kA, (int)BT, kA, 0, g->dec_att_b, nullptr};
    cb.j[3] = ColsumJob{ws.dwf_acc, kA, nlch * B, kA, 0, g->full_att_w, nullptr};
    cb.j[4] = ColsumJob{ws.dbf_acc, 1, nlch * B, 1, 0, g->full_att_b, nullptr};
    cb.j[5] = ColsumJob{ws.dPacc, kA, B * cells, kA, 0, g->enc_att_b, nullptr};
    cb.j[6] = ColsumJob{ws.dinit, 2 * kH, B, 2 * kH, 0, g->init_b, nullptr};
    DIC_TRY(colsum_batch(cb, 7, cs, st));
  }
  // ---- batched weight gradients ---------------------------------------------------------------
  const float* Hprev = ws.Xall + kE + kD;                     // h_{t-1} rows, ld = kXK
  // Five independent products with K-major operands, one launch (gemm_launch_group_colk; until round 4 five launches + three
  // split-K reduces, 0.26 ms of the main stream per step):
  //   [dW_ih | dW_hh] = dG^T [X | h_prev]     f_beta: dgpre^T h_prev     decoder_att: dq^T h_prev     encoder_att: dP^T F
  //   init_linear: dinit^T mean
  {
    GemmParams gp[5] = {};
    auto set = [&](int i, int M, int N, int K, GemmOperand A, GemmOperand Bop, GemmEpilogue ep, int splitk, float* wsp) {
      gp[i].M = M; gp[i].N = N; gp[i].K = K; gp[i].A = A; gp[i].B = Bop; gp[i].ep = ep; gp[i].splitk = splitk; gp[i].ws = wsp;
    };
    GemmEpilogue ep = ep_store(g->w_ih, kE + kD);
    ep.C2 = g->w_hh; ep.ldc2 = kH; ep.nsplit = kE + kD;
    set(0, kG, kXK, (int)BT, op_colk(ws.dG, kG), op_colk(ws.Xall, kXK), ep, 1, nullptr);
    set(1, kD, kH, (int)BT, op_colk(ws.dgpre, kD), op_colk(Hprev, kXK), ep_store(g->fbeta_w, kH), 1, nullptr);
    set(2, kA, kH, (int)BT, op_colk(ws.dq, kA), op_colk(Hprev, kXK), ep_store(g->dec_att_w, kH), 8, ws.gemm_ws);
    set(3, kA, kD, B * cells, op_colk(ws.dPacc, kA), op_colk(ws.F, kD), ep_store(g->enc_att_w, kD), 8, ws.gemm_ws + (size_t)8 * kA * kH);
    set(4, 2 * kH, kD, B, op_colk(ws.dinit, 2 * kH), op_colk(ws.mean, kD), ep_store(g->init_w, kD), 1, nullptr);
    DIC_TRY(gemm_launch_group_colk(gp, 5, st));
    DIC_CHECK_HIP(hipMemcpyAsync(g->b_hh, g->b_ih, sizeof(float) * kG, hipMemcpyDeviceToDevice, st));
  }
  DIC_TRY(gemm(B, kD, 2 * kH, op_rowk(ws.dinit, 2 * kH), op_colk(w->init_w, kD), ep_store(ws.dmean, kD), st, 8,
               ws.gemm_ws, 64));
  // ---- gradient w.r.t. the fused feature map (same for F_rgb and F_depth: F = F_rgb + F_depth) ----
  if (d_features) {
    if (T <= 32) {
      DIC_CELLS_SWITCH(cells, hipLaunchKernelGGL((dF_init_kernel<L_, 32>), dim3(kNCH, B), dim3(256), L_ * 32 * sizeof(float),
                                                 st, alphas, ws.dctx, ws.dmean, T, 0, d_len, d_features);)
    } else {
      DIC_CELLS_SWITCH(cells, hipLaunchKernelGGL((dF_init_kernel<L_, 64>), dim3(kNCH, B), dim3(256), L_ * 64 * sizeof(float),
                                                 st, alphas, ws.dctx, ws.dmean, T, 0, d_len, d_features);)
    }
    DIC_LAUNCH_CHECK();
    GemmEpilogue ep = ep_store(d_features, kD);
    ep.accumulate = 1;
    // dF += dP W_z: W_z^T ([D][A], K-contiguous rows) keeps this 6.6-GFLOP product on the LDS-DMA kernel
    DIC_TRY(launch_transpose(w->enc_att_w, ws.WzT, kA, kD, st));
    DIC_TRY(gemm(B * cells, kD, kA, op_rowk(ws.dPacc, kA), op_rowk(ws.WzT, kA), ep, st));
  }
  return DIC_OK;
}

extern "C" int dic_decoder_bwd(const dic_decoder_weights* w, int V, const int64_t* captions, int cap_stride,
                               const int* dec_lengths, int B, const float* drop_mult, int mode, float temp,
                               const float* dlogits_packed, const float* dalphas, const float* alphas,
                               const dic_decoder_grads* g, float* d_features, void* workspace, size_t workspace_bytes,
                               void* stream) {
  return decoder_bwd_impl(w, V, captions, cap_stride, dec_lengths, B, drop_mult, mode, temp, dlogits_packed, dalphas, alphas,
                          g, d_features, workspace, workspace_bytes, stream, kL);
}

extern "C" int dic_decoder_bwd_cells(const dic_decoder_weights* w, int V, int cells, const int64_t* captions,
                                     int cap_stride, const int* dec_lengths, int B, const float* drop_mult,
                                     const float* dlogits_packed, const float* dalphas, const float* alphas,
                                     const dic_decoder_grads* g, float* d_features, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  return decoder_bwd_impl(w, V, captions, cap_stride, dec_lengths, B, drop_mult, 0, 1.0f, dlogits_packed, dalphas, alphas, g,
                          d_features, workspace, workspace_bytes, stream, cells);
}

size_t dic_decoder_greedy_workspace_bytes(int B, int max_length, int V) {
  bool ov;
  return decoder_carve(nullptr, 0, B, max_length, V, B * max_length, &ov).bytes;
}

int dic_decoder_greedy(const dic_decoder_weights* w, int V, const float* feat_rgb, const float* feat_depth, int B,
                       long long id_start, int max_length, int mode, const float* gumbel_u, int64_t* out_ids,
                       float* alphas_out, void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DIC_TRY(check_common(w, V, B, workspace));
  DIC_REQUIRE(feat_rgb && out_ids && max_length >= 1, "decoder_greedy: bad arguments");
  DIC_REQUIRE(mode == 0 || mode == 2, "decoder_greedy: mode must be 0 (soft) or 2 (Gumbel-max hard attention)");
  DIC_REQUIRE(mode == 0 || gumbel_u != nullptr, "decoder_greedy: hard attention needs the uniform draws");
  const int T = max_length, N = B * T;
  bool ov = false;
  DecoderWs ws = decoder_carve(workspace, workspace_bytes, B, T, V, N, &ov);
  DIC_REQUIRE(!ov, "decoder_greedy: workspace too small (%zu < %zu)", workspace_bytes, ws.bytes);
  float* alphas = alphas_out ? alphas_out : ws.dalp;     // [B,T,196] needed by the step kernel; dalp is [8,B,196]
  if (!alphas_out) DIC_REQUIRE(T <= kNCH, "decoder_greedy: alphas_out required when max_length > %d", kNCH);
  hipLaunchKernelGGL(pack_lstm_kernel, dim3(kG), dim3(256), 0, st, w->w_ih, w->w_hh, w->b_ih, w->b_hh, ws.Wcat, ws.bcat);
  DIC_TRY(launch_transpose(w->dec_att_w, ws.WhT, kA, kH, st));
  DIC_TRY(launch_transpose(w->fbeta_w, ws.WbT, kD, kH, st));
  hipLaunchKernelGGL(fuse_mean_kernel<kL>, dim3(kNCH, B), dim3(256), 0, st, feat_rgb, feat_depth, ws.F, ws.mean);
  DIC_LAUNCH_CHECK();
  DIC_TRY(gemm(B * kL, kA, kD, op_rowk(ws.F, kD), op_rowk(w->enc_att_w, kD), ep_store(ws.P, kA, w->enc_att_b), st));
  {
    GemmEpilogue ep = ep_store(ws.Hall, (long long)(T + 1) * kH, w->init_b);
    ep.C2 = ws.Call; ep.ldc2 = (long long)(T + 1) * kH; ep.nsplit = kH;
    DIC_TRY(gemm(B, 2 * kH, kD, op_rowk(ws.mean, kD), op_rowk(w->init_w, kD), ep, st, 16, ws.gemm_ws, 64));
  }
  hipLaunchKernelGGL(fill_ids_kernel, dim3(ceil_div(B, 256)), dim3(256), 0, st, ws.ids, B, id_start);
  DIC_LAUNCH_CHECK();
  for (int t = 0; t < T; ++t) {
    hipLaunchKernelGGL(embed_step_kernel, dim3(B), dim3(kE), 0, st, w->embed, ws.ids, t, T, V, ws.Xall);
    hipLaunchKernelGGL(attn_fwd_kernel<kL>, dim3(kNCH, (B + 7) / 8 * 8), dim3(512), 0, st, ws.F, ws.P, ws.Hall, ws.WhT, w->dec_att_b,
                       w->full_att_w, w->full_att_b, ws.WbT, w->fbeta_b, t, T, mode, gumbel_u, B, 1.0f, alphas,
                       ws.Qall, ws.ctx, ws.gate, ws.Xall, 1, FusedLstm{}, B);
    DIC_LAUNCH_CHECK();
    DIC_TRY(gemm_slabs(B, kG, kXK, op_rowk(ws.Xall + (long long)t * kXK, (long long)T * kXK), op_rowk(ws.Wcat, kXK),
                       ws.slab_g, kS_LSTM, st));
    hipLaunchKernelGGL(lstm_fwd_kernel, dim3(B), dim3(kH), 0, st, ws.slab_g, kS_LSTM, B, ws.bcat, t, T,
                       (const float*)nullptr, t * B, ws.Hall, ws.Call, ws.Gact, ws.Hdrop);
    DIC_LAUNCH_CHECK();
    // pred = linear(h) (no dropout, depth_models.py:295); softmax is monotone -> argmax of the logits
    DIC_TRY(gemm(B, V, kH, op_rowk(ws.Hdrop + (long long)t * B * kH, kH), op_rowk(w->out_w, kH),
                 ep_store(ws.logits_step, V, w->out_b), st, 1, nullptr, 64));
    hipLaunchKernelGGL(argmax_kernel, dim3(B), dim3(256), 0, st, ws.logits_step, V, t, T, ws.ids, (long long*)out_ids);
    DIC_LAUNCH_CHECK();
  }
  return DIC_OK;
}

size_t dic_attention_workspace_bytes(int B) {
  Carver c(nullptr, 0);
  c.take<float>((size_t)B * kL * kA);
  c.take<float>((size_t)kH * kA);
  c.take<float>((size_t)B * 2 * kH);
  return c.off;
}

int dic_attention_fwd(const float* enc_att_w, const float* enc_att_b, const float* dec_att_w, const float* dec_att_b,
                      const float* full_att_w, const float* full_att_b, const float* feats, const float* h, int B,
                      int mode, const float* gumbel_u, float temp, float* ctx, float* alpha, void* workspace,
                      size_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DIC_REQUIRE(enc_att_w && enc_att_b && dec_att_w && dec_att_b && full_att_w && full_att_b && feats && h && ctx &&
                  alpha && workspace && B > 0, "attention_fwd: bad arguments");
  DIC_REQUIRE(mode >= 0 && mode <= 2 && (mode == 0 || gumbel_u), "attention_fwd: bad mode / missing uniform draws");
  DIC_REQUIRE(workspace_bytes >= dic_attention_workspace_bytes(B), "attention_fwd: workspace too small");
  Carver c(workspace, workspace_bytes);
  float* P = c.take<float>((size_t)B * kL * kA);
  float* WhT = c.take<float>((size_t)kH * kA);
  float* H2 = c.take<float>((size_t)B * 2 * kH);
  DIC_CHECK_HIP(hipMemcpy2DAsync(H2, 2 * kH * sizeof(float), h, kH * sizeof(float), kH * sizeof(float), B,
                                 hipMemcpyDeviceToDevice, st));
  DIC_TRY(launch_transpose(dec_att_w, WhT, kA, kH, st));
  DIC_TRY(gemm(B * kL, kA, kD, op_rowk(feats, kD), op_rowk(enc_att_w, kD), ep_store(P, kA, enc_att_b), st));
  hipLaunchKernelGGL(attn_fwd_kernel<kL>, dim3(kNCH, (B + 7) / 8 * 8), dim3(512), 0, st, feats, (const float*)P, (const float*)H2,
                     (const float*)WhT, dec_att_b, full_att_w, full_att_b, (const float*)nullptr,
                     (const float*)nullptr, 0, 1, mode, gumbel_u, B, temp, alpha, (float*)nullptr, ctx,
                     (float*)nullptr, (float*)nullptr, 0, FusedLstm{}, B);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

size_t dic_attention_bwd_workspace_bytes(int B) {
  Carver c(nullptr, 0);
  c.take<float>((size_t)B * kL * kA); c.take<float>((size_t)B * kL * kA);     // P, dP
  c.take<float>((size_t)B * kA); c.take<float>((size_t)B * kA); c.take<float>((size_t)B);   // dq, dw partials, db partials
  c.take<float>((size_t)kA * kD);                                            // W_z^T
  c.take<float>((size_t)64 * kA);                                            // column-sum scratch
  return c.off;
}

int dic_attention_bwd(const float* enc_att_w, const float* enc_att_b, const float* dec_att_w, const float* dec_att_b,
                      const float* full_att_w, const float* feats, const float* h, const float* alpha, int B, int mode,
                      float temp, const float* d_ctx, const float* d_alpha, float* g_enc_att_w, float* g_enc_att_b,
                      float* g_dec_att_w, float* g_dec_att_b, float* g_full_att_w, float* g_full_att_b, float* d_feats,
                      float* d_h, void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DIC_REQUIRE(enc_att_w && enc_att_b && dec_att_w && dec_att_b && full_att_w && feats && h && alpha && d_ctx &&
                  g_enc_att_w && g_enc_att_b && g_dec_att_w && g_dec_att_b && g_full_att_w && g_full_att_b && d_feats &&
                  d_h && workspace && B > 0, "attention_bwd: bad arguments");
  DIC_REQUIRE(mode == 0 || mode == 1, "attention_bwd: only soft (0) and Gumbel-softmax (1) attention are differentiable");
  DIC_REQUIRE(workspace_bytes >= dic_attention_bwd_workspace_bytes(B), "attention_bwd: workspace too small");
  Carver c(workspace, workspace_bytes);
  float* P = c.take<float>((size_t)B * kL * kA);
  float* dP = c.take<float>((size_t)B * kL * kA);
  float* dq = c.take<float>((size_t)B * kA);
  float* dwp = c.take<float>((size_t)B * kA);
  float* dbp = c.take<float>((size_t)B);
  float* WzT = c.take<float>((size_t)kA * kD);
  float* cs = c.take<float>((size_t)64 * kA);
  DIC_TRY(gemm(B * kL, kA, kD, op_rowk(feats, kD), op_rowk(enc_att_w, kD), ep_store(P, kA, enc_att_b), st));
  hipLaunchKernelGGL(attention_bwd_kernel, dim3(B), dim3(256), 0, st, feats, (const float*)P, h, dec_att_w, dec_att_b,
                     full_att_w, alpha, d_ctx, d_alpha, mode == 1 ? 1.0f / temp : 1.0f, dP, dq, dwp, dbp, d_feats);
  DIC_LAUNCH_CHECK();
  // encoder_att: dW_z = dP^T F, db_z = colsum(dP), dF += dP W_z
  DIC_TRY(gemm(kA, kD, B * kL, op_colk(dP, kA), op_colk(feats, kD), ep_store(g_enc_att_w, kD), st));
  DIC_TRY(colsum(dP, kA, B * kL, kA, g_enc_att_b, cs, st));
  DIC_TRY(launch_transpose(enc_att_w, WzT, kA, kD, st));
  {
    GemmEpilogue ep = ep_store(d_feats, kD);
    ep.accumulate = 1;
    DIC_TRY(gemm(B * kL, kD, kA, op_rowk(dP, kA), op_rowk(WzT, kA), ep, st));
  }
  // decoder_att: dW_h = dq^T h, db_h = colsum(dq), dh = dq W_h
  DIC_TRY(gemm(kA, kH, B, op_colk(dq, kA), op_colk(h, kH), ep_store(g_dec_att_w, kH), st));
  DIC_TRY(colsum(dq, kA, B, kA, g_dec_att_b, cs, st));
  DIC_TRY(gemm(B, kH, kA, op_rowk(dq, kA), op_colk(dec_att_w, kH), ep_store(d_h, kH), st));
  // full_att
  DIC_TRY(colsum(dwp, kA, B, kA, g_full_att_w, cs, st));
  DIC_TRY(colsum(dbp, 1, B, 1, g_full_att_b, cs, st));
  return DIC_OK;
}

}  // extern "C"

#ifdef DIC_EXPERIMENTS
/* development aid (not in dic.h): the phase time stamps of the most recent attn_fwd_kernel launch (2 workgroups x 16 stamps) */
extern "C" int dic_debug_attn_stamps(unsigned long long* host32) {
  DIC_CHECK_HIP(hipMemcpyFromSymbol(host32, HIP_SYMBOL(dic::g_attn_stamps), sizeof(unsigned long long) * 32));
  return 0;
}
#endif
