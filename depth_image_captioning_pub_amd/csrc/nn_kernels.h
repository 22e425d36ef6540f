// BatchNorm / pooling / elementwise kernels shared by the two CNN encoders (NHWC activations).
#pragma once
#include "common.h"

namespace dic {

constexpr float kBnEps = 1e-5f;      // nn.BatchNorm2d defaults
constexpr float kBnMomentum = 0.1f;

// per-layer BatchNorm scratch: scale/shift used by the apply kernels, mean/invstd saved for backward
struct BnBuf {
  float *scale, *shift, *mean, *invstd;
};

// A power-of-two scale of the f16x2 operand format that is CHOSEN ON THE DEVICE (trained weights, gradients: their magnitude is not
// known on the host without a synchronisation).  `bound` receives the bit pattern of an upper bound of |x| (atomicMax of
// non-negative floats = atomicMax of their bits), pow2_scale turns it into s = 2^(13 - floor(log2 bound)) - bound * s in
// [2^13, 2^14): fp16's exponent range leaves room for bounds that are loose by several powers of two - and 1 / s; the kernels that
// split read slot[0], the contraction epilogues read slot[1] through GemmEpilogue::alpha_dev.  All in the caller's workspace.
struct F16Scale {
  unsigned* bound;     // device word, zeroed by f16_scale_reset
  float* slot;         // device {s, 1 / s}
};
int f16_scale_reset(unsigned* bounds, int n, hipStream_t st);                       // bounds[0..n) = 0
int f16_scale_from_absmax(const float* x, long long n, F16Scale s, hipStream_t st); // bound = max |x| (exact), then the slot
int f16_scale_finish(F16Scale s, hipStream_t st);                                   // slot from a bound other kernels have raised

// train mode: reduce the conv epilogue's per-tile partial sums [mtiles][2][C] (fp64), produce
// scale/shift (+ saved mean/invstd) and update the running statistics (unbiased variance).
// `red`: fp64 scratch of bn_finalize_ws_doubles(max mtiles, C) doubles.
void bn_finalize_two_level_rows(int rows);      // partial-sum rows above which bn_finalize_train takes two launches (default 1024; 512 until round 4)
size_t bn_finalize_ws_doubles(long long max_mtiles, int C);
int bn_finalize_train(const float* partial, int mtiles, long long count, int C, const float* gamma,
                      const float* beta, float* running_mean, float* running_var, BnBuf out, double* red,
                      hipStream_t st, unsigned* status = nullptr)   /* status (nullable): raised when a channel's statistics are not finite
                                                                      (an inf / NaN reached the convolution output); that channel's running
                                                                      statistics are then left untouched */;
// eval mode: scale/shift from the running statistics
int bn_finalize_eval(int C, const float* gamma, const float* beta, const float* running_mean,
                     const float* running_var, BnBuf out, hipStream_t st);
// y = act(x*scale[c] + shift[c] (+ residual)); y may alias x. n = rows*C, C % 4 == 0.
int bn_apply(const float* x, const float* residual, float* y, long long rows, int C, BnBuf bn, int relu,
             hipStream_t st);
// same, output as three bf16 planes (hi, mid, lo; exact split of the fp32 value) + optional fp32 copy `y`;
// the residual may be given as fp32 or as planes of the same layout (reconstructed exactly as (hi + mid) + lo)
int bn_apply_planes(const float* x, const float* residual, const unsigned short* const residual_planes[3], float* y,
                    unsigned short* const planes[3], long long rows, int C, BnBuf bn, int relu, hipStream_t st,
                    const BnBuf* residual_bn = nullptr   /* residual_bn: the fp32 residual is raw, apply this affine first */,
                    unsigned* status = nullptr)          /* f16x2 planes (planes[2] == NULL): overflow guard word (common.h) */;
// y[b,ph,pw,c] = max over kxk window (stride s, pad p) of act(x*scale+shift); idx (nullable) = kh*k+kw of the max
int bn_relu_maxpool(const float* x, int B, int H, int W, int C, const BnBuf* bn, int relu, int k, int s, int p,
                    float* y, unsigned char* idx, hipStream_t st,
                    unsigned short* const planes[3] = nullptr   /* optional: also/only paired bf16x3 planes (y may be null then) */,
                    unsigned* status = nullptr                  /* f16x2 planes: overflow guard word */,
                    float* xsel = nullptr                        /* optional [B,PH,PW,C]: the RAW x at each window's argmax (depth_layer1.hip) */);
// y[0..n) = NaN if *status != 0 (the loud end of the f16x2 overflow guard: one small launch, returns at once otherwise)
int poison_if_raised(float* y, long long n, const unsigned* status, hipStream_t st);
int clear_status(unsigned* status, hipStream_t st);      // status[0..63] = 0 (a kernel: see nn_kernels.hip for why not a memset)
// adaptive average pooling of an NHWC map to OUTxOUT (AdaptiveAvgPool2d(14): exact 2x2 replication for 7x7)
// with optional fused BN+ReLU on load
int adaptive_avgpool(const float* x, int B, int H, int W, int C, const BnBuf* bn, int relu, int out, float* y,
                     hipStream_t st);
int adaptive_avgpool_bwd(const float* dy, int B, int H, int W, int C, int out, float* dx, hipStream_t st);

// backward helpers (depth encoder)
// dy[b,h,w,c] = (argmax of window == this pixel ? dpool : 0) * (x*scale+shift > 0)   (k == stride, no padding)
int maxpool_relu_bwd(const float* dpool, const unsigned char* idx, const float* x, int B, int H, int W, int C,
                     int k, BnBuf bn, float* dy, hipStream_t st);
// dy *= (x*scale+shift > 0)
int relu_mask_bwd(float* dy, const float* x, long long rows, int C, BnBuf bn, hipStream_t st);
// BatchNorm backward (train mode): dgamma, dbeta and dx (in place over dy). ws: >= 2*64*C + 3*C floats.
int bn_backward(float* dy_dx, const float* x, long long rows, int C, const float* gamma, BnBuf bn, float* dgamma,
                float* dbeta, float* ws, hipStream_t st,
                unsigned short* const dx_planes[3] = nullptr   /* optional: result also as paired bf16x3 planes (f16x2 planes when [2] == NULL) */,
                F16Scale* f16 = nullptr)                        /* f16x2 planes: the device-resident scale slot this call fills and uses */;
// max-pool (non-overlapping k x k) + ReLU + BatchNorm backward in two passes over x, without materialising the pooled
// gradient (replaces maxpool_relu_bwd + bn_backward); dy receives the gradient w.r.t. the convolution output
int bn_pool_backward(const float* dpool, const unsigned char* idx, const float* x, int B, int H, int W, int C, int k,
                     const float* gamma, BnBuf bn, float* dgamma, float* dbeta, float* ws, float* dy, hipStream_t st,
                     unsigned short* const dy_planes[3] = nullptr   /* optional: dy also as paired bf16x3 planes (f16x2 when [2] == NULL) */,
                     F16Scale* f16 = nullptr)                        /* f16x2 planes: the device-resident scale slot this call fills and uses */;
// (pieces of bn_backward for a caller that forms the partial sums itself - depth_layer1.hip: layout of that workspace, and the finalize
//  [chunks][2][C] partials -> dgamma, dbeta, k2 = mean(g), k3 = mean(g * xhat))
void bn_backward_ws_layout(float* ws, int C, float** part, float** k2, float** k3);
int bn_backward_finalize(const float* part, int chunks, int C, double rows, float* dgamma, float* dbeta, float* k2, float* k3, hipStream_t st);
size_t bn_backward_ws_floats(int C);      // (includes the per-block |g| maxima of the f16x2 scale bound)
// diagnostic: out[r*C + c] = 1 where relu_mask_bwd keeps the gradient (BN output > 0), else 0
int relu_mask_export(const float* x, long long rows, int C, BnBuf bn, unsigned char* out, hipStream_t st);
// column sums (bias gradients): out[c] = sum_r X[r*ld + c]
int colsum_rows(const float* X, long long ld, long long rows, int C, float* out, float* ws, hipStream_t st);
// layout transforms for weights: OIHW <-> OHWI
int oihw_to_ohwi(const float* src, float* dst, int O, int I, int KH, int KW, hipStream_t st);
int ohwi_to_oihw(const float* src, float* dst, int O, int I, int KH, int KW, hipStream_t st);

}  // namespace dic
