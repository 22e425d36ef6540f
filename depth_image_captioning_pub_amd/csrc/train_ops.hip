// Loss, optimiser and RNG kernels of the training step (depth_train.py:210-221).
#include "dic.h"
#include "common.h"
#include <algorithm>
#include <cmath>
#include <vector>

namespace dic {

// ---- cross-entropy forward + backward, one workgroup per packed token ------------------------
// loss_row = logsumexp(x) - x[target]; dlogits = (softmax(x) - onehot) * gscale   (gscale = 1/N)
// `dlogits` MAY ALIAS `logits` (in-place gradient): neither pointer is __restrict__.  (They were in round 1: the compiler
// was then free to sink thread 0's read of x[target] below the barrier, where another thread may already have stored
// that element's gradient - the row's loss, and only the loss, came out wrong whenever the timing allowed it.)
__global__ void __launch_bounds__(256) ce_fwd_bwd_kernel(const float* logits,
                                                          const long long* __restrict__ targets, int V, float gscale,
                                                          float* __restrict__ loss_rows, float* dlogits) {
  __shared__ float red[8];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* x = logits + (long long)row * V;
  float m = -INFINITY;
  for (int v = tid; v < V; v += 256) m = fmaxf(m, x[v]);
  m = wave_max(m);
  if (lane == 0) red[w] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.f;
  for (int v = tid; v < V; v += 256) s += expf(x[v] - m);
  s = wave_sum(s);
  if (lane == 0) red[4 + w] = s;
  __syncthreads();
  s = red[4] + red[5] + red[6] + red[7];
  const float lse = m + logf(s);
  long long tg = targets[row];
  // F.cross_entropy raises on a class index outside [0, V) (depth_train.py:214); a kernel cannot raise, so the row's
  // loss becomes NaN - the step's loss is then NaN, which no caller can mistake for a result - and the address is clamped
  const bool bad_target = tg < 0 || tg >= V;
  tg = tg < 0 ? 0 : (tg >= V ? V - 1 : tg);
  if (tid == 0) loss_rows[row] = bad_target ? __builtin_nanf("") : lse - x[tg];
  __threadfence_block();
  __syncthreads();                       // dlogits may alias logits: the read of x[tg] is done before any store below
  float* dx = dlogits + (long long)row * V;
  for (int v = tid; v < V; v += 256) {
    const float p = expf(x[v] - lse);
    dx[v] = (p - (v == tg ? 1.f : 0.f)) * gscale;
  }
}

// ---- doubly-stochastic attention regulariser: r[b,l] = 1 - sum_t alpha[b,t,l] -------------------
__global__ void __launch_bounds__(256) alpha_reg_kernel(const float* __restrict__ alphas, int T, float coef,
                                                         float* __restrict__ reg_rows, float* __restrict__ dalphas) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  float r = 0.f;
  if (tid < DIC_L) {
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += alphas[((long long)b * T + t) * DIC_L + tid];
    r = 1.f - s;
    const float g = -2.f * r * coef;                 // d/d alpha of coef * r^2
    for (int t = 0; t < T; ++t) dalphas[((long long)b * T + t) * DIC_L + tid] = g;
  }
  const float p = wave_sum(r * r);
  if (lane == 0) red[w] = p;
  __syncthreads();
  if (tid == 0) reg_rows[b] = red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256) loss_finish_kernel(const float* __restrict__ loss_rows, int N,
                                                           const float* __restrict__ reg_rows, int B, float reg_scale,
                                                           float* __restrict__ loss) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < N; i += 256) s += (double)loss_rows[i];
  double r = 0.0;
  if (reg_rows)
    for (int i = threadIdx.x; i < B; i += 256) r += (double)reg_rows[i];
  red[threadIdx.x] = s / (double)N + r * (double)reg_scale;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)red[0];
}

// one workgroup per decode step t: rows with dec_len > t are a prefix (lengths sorted descending), their packed offset
// is sum_{t' < t} bs[t'] = sum_b min(dec_len[b], t).  Both counts are recomputed per workgroup from the device copy of
// the lengths, so no host-built table has to be staged.
__global__ void __launch_bounds__(256) pack_targets_kernel(const long long* __restrict__ cap, int cap_stride,
                                                            const int* __restrict__ dec_len, int B,
                                                            long long* __restrict__ out) {
  __shared__ int red[2][4];
  const int t = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int off = 0, nb = 0;
  for (int b = threadIdx.x; b < B; b += 256) {
    const int l = dec_len[b];
    off += min(l, t);
    nb += l > t;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { off += __shfl_xor(off, o, 64); nb += __shfl_xor(nb, o, 64); }
  if (lane == 0) { red[0][w] = off; red[1][w] = nb; }
  __syncthreads();
  off = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  nb = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  for (int b = threadIdx.x; b < nb; b += 256) out[off + b] = cap[(long long)b * cap_stride + t + 1];
}

// ---- AdamW (torch.optim.AdamW single-tensor math, fp32) -----------------------------------------
__global__ void __launch_bounds__(256) adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                     float* __restrict__ m, float* __restrict__ v, long long n,
                                                     float decay, float beta1, float beta2, float step_size,
                                                     float inv_bc2_sqrt, float eps, const unsigned* __restrict__ skip_if_raised) {
  if (skip_if_raised && *skip_if_raised != 0u) return;       // f16x2 overflow guard (dic.h): parameters and moments stay as they are
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float gi = g[i];
    float pi = p[i] * decay;                                    // p *= 1 - lr*wd
    const float mi = m[i] + (gi - m[i]) * (1.f - beta1);        // exp_avg.lerp_(grad, 1-beta1)
    const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;    // exp_avg_sq.mul_(b2).addcmul_(g,g,1-b2)
    const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
    pi -= step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

// ---- Philox4x32-10 dropout multiplier ---------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__global__ void __launch_bounds__(256) dropout_mask_kernel(float* __restrict__ out, long long n, float p, float scale,
                                                            uint64_t seed, uint64_t offset) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;      // one Philox block -> 4 outputs
  if (q * 4 >= n) return;
  const uint64_t ctr = (uint64_t)q + offset;
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long i = q * 4 + j;
    if (i < n) {
      const float u = (float)(c[j] >> 8) * (1.0f / 16777216.0f);     // [0,1)
      out[i] = (u >= p) ? scale : 0.f;
    }
  }
}

}  // namespace dic

using namespace dic;

extern "C" {

int dic_caption_loss(const float* logits, const int64_t* targets, int n_packed, int V, const float* alphas, int B,
                     int Tmax, float lam, float ce_grad_scale, float reg_grad_scale, float* loss, float* dlogits,
                     float* dalphas, float* scratch, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DIC_REQUIRE(logits && targets && loss && dlogits && scratch, "caption_loss: null pointer");
  DIC_REQUIRE(n_packed > 0 && V > 0, "caption_loss: empty input");
  float* loss_rows = scratch;
  float* reg_rows = scratch + n_packed;
  hipLaunchKernelGGL(ce_fwd_bwd_kernel, dim3(n_packed), dim3(256), 0, st, logits, (const long long*)targets, V,
                     ce_grad_scale / (float)n_packed, loss_rows, dlogits);
  DIC_LAUNCH_CHECK();
  float reg_scale = 0.f;
  if (alphas) {
    DIC_REQUIRE(dalphas != nullptr && B > 0 && Tmax > 0, "caption_loss: dalphas required with alphas");
    reg_scale = lam / ((float)B * (float)DIC_L);
    hipLaunchKernelGGL(alpha_reg_kernel, dim3(B), dim3(256), 0, st, alphas, Tmax, reg_scale * reg_grad_scale, reg_rows,
                       dalphas);
    DIC_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, st, loss_rows, n_packed,
                     alphas ? reg_rows : (const float*)nullptr, B, reg_scale, loss);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

int dic_pack_targets(const int64_t* captions, int cap_stride, const int* dec_lengths, int B, int64_t* targets,
                     void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DIC_REQUIRE(captions && dec_lengths && targets && B > 0, "pack_targets: bad arguments");
  const int T = dec_lengths[0];
  DIC_REQUIRE(T >= 1 && T < 4096, "pack_targets: bad length");
  long long off = 0;
  for (int b = 0; b < B; ++b) {
    DIC_REQUIRE(dec_lengths[b] >= 1 && (b == 0 || dec_lengths[b] <= dec_lengths[b - 1]),
                "pack_targets: lengths must be >= 1 and sorted in descending order");
    off += dec_lengths[b];
  }
  // the caller's `dec_lengths` array is only read during this call: hipMemcpyAsync from pageable host memory returns
  // after the bytes have been staged, and nothing function-local is handed to the copy engine
  int* d_len = reinterpret_cast<int*>(targets + off);       // device copy of the lengths lives in the tail (include/dic.h)
  DIC_CHECK_HIP(hipMemcpyAsync(d_len, dec_lengths, sizeof(int) * B, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(pack_targets_kernel, dim3(T), dim3(256), 0, st, (const long long*)captions, cap_stride, d_len, B,
                     (long long*)targets);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

int dic_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, int step,
                   float lr, float beta1, float beta2, float eps, float weight_decay, void* stream) {
  return dic_adamw_step_guarded(params, grads, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, weight_decay, nullptr, stream);
}

int dic_adamw_step_guarded(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, int step,
                           float lr, float beta1, float beta2, float eps, float weight_decay, const uint32_t* skip_if_raised,
                           void* stream) {
  DIC_REQUIRE(params && grads && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adamw: bad arguments");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const int blocks = (int)std::min<long long>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq,
                     n, 1.0f - lr * weight_decay, beta1, beta2, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), eps,
                     (const unsigned*)skip_if_raised);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

int dic_dropout_mask(float* out, long long n, float p, uint64_t seed, uint64_t offset, void* stream) {
  DIC_REQUIRE(out && n > 0 && p >= 0.f && p < 1.f, "dropout_mask: bad arguments");
  const long long q = (n + 3) / 4;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, n,
                     p, 1.0f / (1.0f - p), seed, offset);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

}  // extern "C"
