// extern "C" surface of libdic_hip.so (declared in include/dic.h).
#include "dic.h"
#include "conv.h"

using namespace dic;

namespace dic { void resnet_fuse_bn_operand(int mask); void resnet_fuse_bn_halo(int on); void resnet_fuse_res_bn(int on); void depth_encoder_l1_sparse(int on); void bn_finalize_two_level_rows(int rows); void resnet_debug_fused_tail_bn(int on); void conv1_depth_debug_blocks(int n);
                void depth_encoder_f16x2(int on); }
#ifdef DIC_EXPERIMENTS
namespace dic { void decoder_debug_persistent(int on); void decoder_persist_debug_buffer(unsigned long long* p); void decoder_persist_debug_placement(int p);
                void resnet_debug_skip_bn_apply(int on); }
#endif

extern "C" {

int dic_version(void) { return DIC_ABI_VERSION; }
size_t dic_struct_bytes(int which) {
  switch (which) {
    case 0: return sizeof(dic_conv_bn_layer);
    case 1: return sizeof(dic_decoder_weights);
    case 2: return sizeof(dic_decoder_grads);
    case 3: return sizeof(dic_depth_encoder_weights);
    case 4: return sizeof(dic_depth_encoder_grads);
    case 5: return sizeof(dic_depth_bn_state);
    default: return 0;
  }
}
const char* dic_last_error(void) { return last_error(); }

int dic_gemm_f32(int M, int N, int K, const float* A, long long lda, int a_colk, const float* B, long long ldb,
                 int b_colk, float* C, long long ldc, const float* bias, int act, int accumulate, int splitk,
                 float* workspace, size_t workspace_bytes, int force_tile, void* stream) {
  GemmParams p{};
  p.M = M; p.N = N; p.K = K;
  p.A = a_colk ? op_colk(A, lda) : op_rowk(A, lda);
  p.B = b_colk ? op_colk(B, ldb) : op_rowk(B, ldb);
  p.ep = ep_store(C, ldc, bias, act);
  p.ep.accumulate = accumulate;
  p.splitk = splitk; p.ws = workspace;
  p.ablate = force_tile / 1000;            // benchmark-only ablation switch (scripts/bench_gemm.py)
  force_tile %= 1000;
  DIC_REQUIRE(force_tile == 0 || force_tile == 64 || force_tile == 128, "force_tile must be 0, 64 or 128");
  if (splitk > 1)
    DIC_REQUIRE(workspace_bytes >= gemm_splitk_ws_bytes(M, N, splitk), "dic_gemm_f32: workspace too small");
  return gemm_launch(p, (hipStream_t)stream, force_tile);
}

int dic_conv2d_fwd(const float* x, int B, int H, int W, int C, int in_nchw, const float* w_ohwi, const float* bias,
                   int CO, int KH, int KW, int stride, int pad, float* y_nhwc, float* bn_partial, int* mtiles_out,
                   int force_tile, float* tail_ws, void* stream) {
  ConvDesc d{B, H, W, C, CO, KH, KW, stride, pad, in_nchw};
  return conv_fwd(x, d, w_ohwi, bias, y_nhwc, bn_partial, mtiles_out, (hipStream_t)stream, force_tile, tail_ws);
}

/* Kernel-selection switches (include/dic.h).  The product library accepts the codes its own tests use to put two product
 * kernels side by side; every other code (ablations, parked kernels) exists only in the experiments build. */
int dic_debug_force_staged_gemm(int on) {
  if (gemm_bf3_force_tile(on) == 0) return 0;
  if (on >= 100 && on <= 104) { dic::resnet_fuse_bn_operand(on == 104 ? -1 : on - 100); return 0; }
  if (on == 98 || on == 99) { dic::resnet_fuse_res_bn(on - 98); return 0; }           // downsample-branch BatchNorm inside the on-the-fly 1x1 kernel: never / yes (default)
  if (on == 108 || on == 109) { dic::resnet_fuse_bn_halo(on - 108); return 0; }       // conv1 -> conv2 BatchNorm-apply inside the halo kernel: never / by shape (default)
  if (on == 182 || on == 183) { dic::bn_finalize_two_level_rows(on == 182 ? 512 : 1024); return 0; }   // BatchNorm finalize in two launches above 512 / 1024 (default) rows of partials
  if (on == 180 || on == 181) { dic::depth_encoder_l1_sparse(on - 180); return 0; }   // depth encoder layer-1 backward: dense / sparse (default)
  if (on == 116 || on == 117) { dic::depth_encoder_f16x2(on - 116); return 0; }      // depth encoder conv2 / conv3: bf16x3 / f16x2 (default)
#ifdef DIC_EXPERIMENTS
  if (on == 140 || on == 141) { dic::decoder_debug_persistent(on - 140); return 0; }          // decoder forward: per-step launches / persistent loop
  if (on == 142 || on == 143) { dic::decoder_persist_debug_placement(on - 142); return 0; }   // persistent loop: workgroup placement
  if (on >= 130 && on <= 134) { dic::conv1_depth_debug_blocks(256 * (on - 130)); return 0; }   // generic path / 256 / 512 / 768 / 1024 workgroups
  if (on >= 120 && on <= 123) { dic::resnet_debug_fused_tail_bn(on - 120); return 0; }
  if (on >= 124 && on <= 127) { dic::resnet_debug_skip_bn_apply(on == 125 ? 3 : on == 126 ? 2 : on == 127 ? 1 : 0); return 0; }   // (measurement only) bn_apply_planes: run / skip all / skip block outputs / skip c1, c2 outputs
  if (on >= 0 && on <= 13) { gemm_force_v1(on); return 0; }
#endif
  DIC_REQUIRE(false, "debug switch: unknown code %d", on);
  return 0;
}
#ifdef DIC_EXPERIMENTS
/* development aid (not in dic.h): device buffer [T][8] receiving phase time stamps of the persistent decoder loop */
int dic_debug_decoder_stamps(unsigned long long* dev_buf) { dic::decoder_persist_debug_buffer(dev_buf); return 0; }
#endif
/* development aid (not in dic.h): depth-encoder layer-1 kernels alone (csrc/conv1_depth.hip); y [B,OH,OW,128]; ws >= 1024 * 6400 floats */
int dic_debug_conv1_wgrad(const float* x, int B, int H, int W, const float* dy, float* dw, float* dbias, float* ws, float* cs_ws,
                          void* stream) {
  ConvDesc d{B, H, W, 1, 128, 7, 7, 3, 0, 0};
  return conv1_depth_wgrad(x, d, dy, dw, dbias, ws, cs_ws, (hipStream_t)stream);
}
int dic_debug_conv1_fwd(const float* x, int B, int H, int W, const float* w, const float* bias, float* y, float* partial,
                        void* stream) {
  ConvDesc d{B, H, W, 1, 128, 7, 7, 3, 0, 0};
  int rows = 0;
  return conv1_depth_fwd(x, d, w, bias, y, partial, &rows, (hipStream_t)stream);
}
/* development aid (not in dic.h): one split-bf16 convolution with train-mode BatchNorm partial sums, from paired planes
 * (dic_split_bf16x3_paired of the NHWC input viewed as [B*H*W][C] and of the OHWI weight viewed as [CO][KH*KW*C]) */
int dic_debug_conv_bf3(const uint16_t* const x_planes[3], int B, int H, int W, int C, const uint16_t* const w_planes[3], int CO,
                       int k, int stride, int pad, float* y, float* bn_partial, int* mtiles_out, float* tail_ws, void* stream) {
  ConvDesc d{B, H, W, C, CO, k, k, stride, pad, 0};
  return conv_fwd_bf3(x_planes, d, w_planes, y, bn_partial, mtiles_out, tail_ws, (hipStream_t)stream, nullptr, nullptr, nullptr);
}
/* the same two with the operand format explicit (0 = bf16x3, 1 = f16x2: two planes per operand from dic_split_f16x2_paired,
 * out_scale = 1 / (scale of the x planes * scale of the w planes); the on-the-fly operand is scaled by 4 inside the kernel;
 * tail_ws: 1024 * 64 * 64 floats) */
int dic_debug_conv_fmt(const uint16_t* const x_planes[3], int B, int H, int W, int C, const uint16_t* const w_planes[3], int CO,
                       int k, int stride, int pad, float* y, float* bn_partial, int* mtiles_out, float* tail_ws, int fmt, float out_scale,
                       void* stream) {
  ConvDesc d{B, H, W, C, CO, k, k, stride, pad, 0};
  return conv_fwd_bf3(x_planes, d, w_planes, y, bn_partial, mtiles_out, tail_ws, (hipStream_t)stream, nullptr, nullptr, nullptr, 0, 1024, fmt,
                      out_scale);      // (tail_ws: 1024 slabs of [64][64] floats, as inside the ResNet workspace)
}
int dic_debug_conv1x1_bn_fmt(const float* raw, const float* scale, const float* shift, const float* res, int relu, float* act_out, int M,
                             int C, const uint16_t* const w_planes[3], int CO, float* y, float* bn_partial, int* mtiles_out, float* tail_ws,
                             int tail_ws_slabs, int fmt, float out_scale, void* stream) {
  return conv1x1_fwd_bf3_bn(raw, scale, shift, res, relu, act_out, M, C, w_planes, CO, y, bn_partial, mtiles_out, tail_ws, tail_ws_slabs,
                            (hipStream_t)stream, nullptr, nullptr, fmt, out_scale);
}
/* development aid (not in dic.h): the 3x3 / stride 1 / pad 1 convolution of 14x14 maps with the BatchNorm-apply + ReLU + f16x2 split of its
 * input done inside the LDS-halo kernel (conv3x3_fwd_bf3_bn, gemm_bf3.hip); returns 1 when that kernel does not take the shape */
int dic_debug_conv3x3_bn(const float* raw, const float* scale, const float* shift, int relu, int B, int H, int W, int C,
                         const uint16_t* const w_planes[3], int CO, float* y, float* bn_partial, int* mtiles_out, float* tail_ws,
                         int tail_ws_slabs, float out_scale, uint32_t* status, void* stream) {
  ConvDesc d{B, H, W, C, CO, 3, 3, 1, 1, 0};
  return conv3x3_fwd_bf3_bn(raw, scale, shift, relu, d, w_planes, y, bn_partial, mtiles_out, tail_ws, tail_ws_slabs, (hipStream_t)stream,
                            nullptr, nullptr, 1, out_scale, status);
}
#ifdef DIC_EXPERIMENTS
/* development aid (not in dic.h): the A-stationary conv3 kernel alone (conv1x1_astat_bn, gemm_bf3.hip): y = relu(raw * scale + shift) . W^T
 * with W as f16x2 planes scaled by 1 / (4 * out_scale); bn_partial: (2 * ceil(M / 64)) x 2 x CO floats; returns 1 for shapes it does not take */
int dic_debug_conv1x1_astat(const float* raw, const float* scale, const float* shift, int relu, int M, int C, const uint16_t* const w_planes[3],
                            int CO, float* y, float* bn_partial, int* mtiles_out, float out_scale, uint32_t* status, void* stream) {
  return conv1x1_astat_bn(raw, scale, shift, relu, M, C, w_planes, CO, y, bn_partial, mtiles_out, (hipStream_t)stream, out_scale, status);
}
#endif
int dic_conv_persistent_grid(int max_workgroups) {
  DIC_REQUIRE(gemm_bf3_set_persist_grid(max_workgroups) == 0, "dic_conv_persistent_grid: 1 <= max_workgroups <= 1024");
  return DIC_OK;
}
/* development aid (not in dic.h): the 1x1 convolution with on-the-fly BatchNorm / residual / ReLU / split of its input
 * (conv1x1_fwd_bf3_bn, gemm_bf3.hip); returns 1 when the policy would not run the shape on the persistent kernel */
int dic_debug_conv1x1_bn(const float* raw, const float* scale, const float* shift, const float* res, int relu, float* act_out, int M,
                         int C, const uint16_t* const w_planes[3], int CO, float* y, float* bn_partial, int* mtiles_out, float* tail_ws,
                         int tail_ws_slabs, void* stream) {
  return conv1x1_fwd_bf3_bn(raw, scale, shift, res, relu, act_out, M, C, w_planes, CO, y, bn_partial, mtiles_out, tail_ws, tail_ws_slabs,
                            (hipStream_t)stream, nullptr, nullptr);
}
int dic_profile_begin(void) { return gemm_profile_begin(); }
int dic_profile_end(int max_entries, int* keys, double* total_ms, double* total_flops, long long* launches, int* n_out) {
  DIC_REQUIRE(keys && total_ms && total_flops && launches && n_out && max_entries > 0, "profile_end: bad arguments");
  return gemm_profile_end(max_entries, keys, total_ms, total_flops, launches, n_out);
}
int dic_profile_end_bytes(int max_entries, int* keys, double* total_ms, double* total_flops, double* total_bytes, long long* launches,
                          int* n_out) {
  DIC_REQUIRE(keys && total_ms && total_flops && total_bytes && launches && n_out && max_entries > 0, "profile_end_bytes: bad arguments");
  return gemm_profile_end(max_entries, keys, total_ms, total_flops, launches, n_out, total_bytes);
}

}  // extern "C"
