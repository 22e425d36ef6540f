// Convolution (NHWC activations, OHWI weights) on the generic MFMA contraction kernel.
#pragma once
#include "gemm.h"

namespace dic {

struct ConvDesc {
  int B, H, W, C;            // input
  int CO, KH, KW, stride, pad;
  int in_nchw;               // input tensor is NCHW (first layers only; uses the gather loader)
  int OH() const { return (H + 2 * pad - KH) / stride + 1; }
  int OW() const { return (W + 2 * pad - KW) / stride + 1; }
  int M() const { return B * OH() * OW(); }
  int K() const { return KH * KW * C; }
  ConvGeom geom() const { return ConvGeom{H, W, C, OH(), OW(), KH, KW, stride, pad, in_nchw}; }
};

// y[B,OH,OW,CO] = conv(x, w) (+bias).  bn_partial (nullable): [mtiles][2][CO] column partials.
// tail_ws (nullable): kGemmTailWsBytes of scratch enabling the remainder-tile K split (see gemm.hip).
int conv_fwd(const float* x, const ConvDesc& d, const float* w_ohwi, const float* bias, float* y,
             float* bn_partial, int* mtiles_out, hipStream_t st, int force_tile = 0, float* tail_ws = nullptr);
int conv_mtiles(const ConvDesc& d, int force_tile = 0);

// fp32-accurate convolution on the bf16 matrix cores (gemm_bf3.hip): activations ([B*H*W rows][C]) and weights
// ([CO rows][KH*KW*C]) are given as three bf16 planes (hi, mid, lo) each, in the row-pair interleaved layout
// (plane_offset(..., paired = 1), gemm.h); output is raw fp32 + the same BN partial sums as conv_fwd (64-row M tiles).
int conv_fwd_bf3(const unsigned short* const x_planes[3], const ConvDesc& d, const unsigned short* const w_planes[3],
                 float* y, float* bn_partial, int* mtiles_out, float* tail_ws, hipStream_t st,
                 const float* bias = nullptr, const BnFuseArgs* bn_fuse = nullptr, int* bn_fused = nullptr,
                 int act = 0 /* ACT_NONE */, int tail_ws_slabs = 256 /* [64][64]-float slabs in tail_ws (kGemmTailWsBytes = 256) */,
                 int fmt = 0 /* operand format: 0 = bf16x3, 1 = f16x2 (two fp16 planes of scaled values, gemm_bf3.hip) */,
                 float out_scale = 1.0f /* f16x2: 1 / (scale of the x planes * scale of the w planes) */,
                 const float* alpha_dev0 = nullptr, const float* alpha_dev1 = nullptr /* f16x2: further factors of the result scale that live
                                                                                         in device memory (GemmEpilogue::alpha_dev) */);
// 1x1 convolution with the BatchNorm-apply (+ residual) + ReLU + split of its input fused into the operand path (gemm_bf3.hip);
// returns 1 (nothing launched) when the shape would not run on the persistent warp-specialised kernel
int conv1x1_fwd_bf3_bn(const float* raw, const float* scale, const float* shift, const float* res, int relu, float* act_out,
                       int M, int C, const unsigned short* const w_planes[3], int CO, float* y, float* bn_partial,
                       int* mtiles_out, float* tail_ws, int tail_ws_slabs, hipStream_t st, const BnFuseArgs* bn_fuse = nullptr,
                       int* bn_fused = nullptr, int fmt = 0, float out_scale = 1.0f,
                       unsigned* status = nullptr /* f16x2: overflow guard word (common.h) */,
                       const float* res_scale = nullptr, const float* res_shift = nullptr /* f16x2: BatchNorm of the residual itself */);
bool conv1x1_bf3_bn_eligible(int M, int C, int CO, int tail_ws_slabs, int fmt = 0);
// 3x3 / stride 1 / pad 1 convolution of 14x14 maps with the same fusion, in the LDS-halo kernel's producer waves (f16x2 format only);
// returns 1 (nothing launched) for every other shape / format
int conv3x3_fwd_bf3_bn(const float* raw, const float* scale, const float* shift, int relu, const ConvDesc& d,
                       const unsigned short* const w_planes[3], float* y, float* bn_partial, int* mtiles_out, float* tail_ws,
                       int tail_ws_slabs, hipStream_t st, const BnFuseArgs* bn_fuse, int* bn_fused, int fmt, float out_scale, unsigned* status);
// (parked, experiments build: the same operation for SHORT contractions on the A-stationary kernel - csrc/experiments/conv1x1_astat.inc)
#ifdef DIC_EXPERIMENTS
bool conv1x1_astat_eligible(int M, int C, int CO);
int conv1x1_astat_bn(const float* raw, const float* scale, const float* shift, int relu, int M, int C, const unsigned short* const w_planes[3],
                     int CO, float* y, float* bn_partial, int* mtiles_out, hipStream_t st, float out_scale, unsigned* status = nullptr);
#endif
constexpr int kResnetTailSlabs = 1024;      // the ResNet workspace carves a larger tail region: every CU can take a remainder piece
// bn_fuse: when the launch is tail-split, finalize the train-mode BatchNorm inside the fix-up launch (*bn_fused = 1)
// data gradient of a stride-1 convolution through the same kernel: dy planes [B,OH,OW,CO], flipped weights
// [C][KH][KW][CO] planes (conv_flip_weights + split), dx fp32 [B,H,W,C]
// weight gradient through the same kernel (split-K over the output pixels); dyT / pT: scratch planes of
// conv_wgrad_bf3_plane_elems(d, 0 / 1) elements each, ws: conv_wgrad_bf3_ws_floats(d, splitk) floats
size_t conv_wgrad_bf3_plane_elems(const ConvDesc& d, int which);
size_t conv_wgrad_bf3_ws_floats(const ConvDesc& d, int splitk);
int conv_wgrad_bf3(const float* x, const ConvDesc& d, const float* dy, float* dw_ohwi, int splitk,
                   unsigned short* const dyT[3], unsigned short* const pT[3], float* ws, hipStream_t st,
                   int fmt = 0, const float* dy_slot = nullptr /* f16x2: the gradient's device-resident {scale, 1 / scale} (F16Scale::slot) */);
// 7x7 stride-2 pad-3 stem with C_in = 3 on the bf16x3 kernel (strip formulation, see gemm_bf3.hip)
size_t conv_stem_bf3_plane_elems(int B, int H, int W);
int conv_stem_pack_weights(const float* w_oihw, int CO, float* scratch_f32, unsigned short* const w_planes[3], hipStream_t st,
                           float f16_scale = 0.f /* > 0: two f16x2 planes of f16_scale * w instead of three bf16 planes */);
int conv_stem_bf3(const float* imgs_nchw, int B, int H, int W, int CO, unsigned short* const x_planes[3],
                  const unsigned short* const w_planes[3], float* y, float* bn_partial, int* mtiles_out, hipStream_t st,
                  int fmt = 0, float out_scale = 1.0f, unsigned* status = nullptr /* f16x2: image planes of 4 * x, guarded */);
int conv_dgrad_s1_bf3(const unsigned short* const dy_planes[3], const ConvDesc& d,
                      const unsigned short* const wflip_planes[3], float* dx, hipStream_t st, float* tail_ws = nullptr,
                      int tail_ws_slabs = 256, int fmt = 0, const float* alpha_dev0 = nullptr, const float* alpha_dev1 = nullptr);
int split_bf16x3(const float* x, long long n, unsigned short* hi, unsigned short* mid, unsigned short* lo, hipStream_t st);
int split_bf16x3_paired(const float* x, long long rows, int K, unsigned short* hi, unsigned short* mid,
                        unsigned short* lo, hipStream_t st);
int split_f16x2_paired(const float* x, long long rows, int K, float scale, unsigned short* h1, unsigned short* h2, hipStream_t st,
                       unsigned* status = nullptr /* overflow guard word (common.h) */);

// Depth-encoder layer 1 (1 -> 128 channels, 7x7, stride 3, no padding) on packed fp32 vector FMAs (conv1_depth.hip).
//   fwd: y [B,OH,OW,128] = conv(x [B,H,W]) + bias; bn_partial (nullable) receives one [2][128] row of sums per workgroup,
//        *partial_rows = conv1_depth_fwd_blocks(d) of them
//   wgrad: dw [128][49] (OIHW) and dbias [128] (nullable) from dy [B,OH,OW,128]; ws >= conv1_depth_wgrad_ws_floats(d)
//          floats, cs_ws = colsum_rows scratch (>= 64 * 6272 floats)
int conv1_depth_fwd(const float* x, const ConvDesc& d, const float* w, const float* bias, float* y, float* bn_partial,
                    int* partial_rows, hipStream_t st);
int conv1_depth_fwd_blocks(const ConvDesc& d);
bool conv1_depth_supported(const ConvDesc& d);   // shape fits (rows <= 640 floats wide) and not switched off (debug code 130);
                                                  // otherwise the caller uses the generic gather kernels
size_t conv1_depth_wgrad_ws_floats(const ConvDesc& d);
// the whole backward of that layer (BatchNorm + ReLU + max-pool 3 behind the convolution) without the full-size gradient: depth_layer1.hip
struct BnBuf;
bool depth_layer1_sparse_supported(const ConvDesc& d);
size_t depth_layer1_sparse_ws_floats(const ConvDesc& d);
int conv1_depth_wgrad(const float* x, const ConvDesc& d, const float* dy, float* dw, float* dbias, float* ws, float* cs_ws,
                      hipStream_t st);
// dW[CO][KH][KW][C] = sum_m dY[m,co] * patch(m)[kh,kw,c]   (split over M with workspace `ws`)
int conv_wgrad(const float* x, const ConvDesc& d, const float* dy, float* dw_ohwi, int splitk, float* ws,
               hipStream_t st);
// dX[B,H,W,C] for stride-1 convolutions: full correlation of dY with the flipped weights
// w_flip[C][KH][KW][CO] (see flip_weights_kernel); pad_dgrad = KH-1-pad.
int conv_dgrad_s1(const float* dy, const ConvDesc& d, const float* w_flip, float* dx, hipStream_t st);
int conv_flip_weights(const float* w_ohwi, const ConvDesc& d, float* w_flip, hipStream_t st);

}  // namespace dic
