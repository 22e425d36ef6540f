// The two CNN encoders of the depth-soft captioner on MI355X (NHWC activations, exact-fp32 MFMA
// implicit-GEMM convolutions, train-mode BatchNorm statistics from the conv epilogue).
//
//   depth encoder : Depth_CNN_endoder (Depth_caption_model/depth_models.py:12-56), forward + backward
//   RGB encoder   : CNNEncoder_Atten = torchvision ResNet-152 children()[:-1] with
//                   AdaptiveAvgPool2d(14) (Base_caption_model/base_caption_models.py:18-45), forward
//                   only (@torch.no_grad), BatchNorm in batch-statistics mode when train_bn (quirk Q1).
#include "dic.h"
#include "conv.h"
#include "nn_kernels.h"
#include <algorithm>
#include <vector>

namespace dic {

// ------------------------------------------------------------------------------------------
// depth encoder
// ------------------------------------------------------------------------------------------
struct DepthGeom {
  int B, H, W;
  int H1, W1, P1h, P1w, H2, W2, P2h, P2w;
  ConvDesc c1, c2, c3;
  long long M1, M2, M3;
};

static DepthGeom depth_geom(int B, int H, int W) {
  DepthGeom g{};
  g.B = B; g.H = H; g.W = W;
  g.c1 = ConvDesc{B, H, W, 1, 128, 7, 7, 3, 0, 1};
  g.H1 = g.c1.OH(); g.W1 = g.c1.OW();
  g.P1h = g.H1 / 3; g.P1w = g.W1 / 3;
  g.c2 = ConvDesc{B, g.P1h, g.P1w, 128, 512, 3, 3, 1, 0, 0};
  g.H2 = g.c2.OH(); g.W2 = g.c2.OW();
  g.P2h = g.H2 / 3; g.P2w = g.W2 / 3;
  g.c3 = ConvDesc{B, g.P2h, g.P2w, 512, 2048, 1, 1, 1, 0, 0};
  g.M1 = (long long)B * g.H1 * g.W1; g.M2 = (long long)B * g.H2 * g.W2; g.M3 = (long long)B * g.P2h * g.P2w;
  return g;
}

static int g_l1_sparse = 1;        // codes 180 / 181: backward of the depth encoder's first layer: dense passes over its 175-MB map / sparse (default)
void depth_encoder_l1_sparse(int on) { g_l1_sparse = on; }
int depth_layer1_backward_sparse(const float* depth, const ConvDesc& d, const float* dpool, const unsigned char* idx, const float* xsel,
                                 const float* conv_w, const float* conv_b, const float* gamma, BnBuf bn, float* dgamma, float* dbeta,
                                 float* dW, float* db, float* bn_ws, float* ws, float* cs_ws, hipStream_t st);      // depth_layer1.hip
static int g_depth_f16x2 = 1;      // codes 116 / 117: conv2 / conv3 of the depth encoder in bf16x3 / f16x2 (default) arithmetic
void depth_encoder_f16x2(int on) { g_depth_f16x2 = on; }
constexpr int kWg1Split = 128;   // split-K of the conv1 weight gradient (K = B*73*73)
constexpr int kWg2Split = 4;
constexpr int kWg2SplitBf3 = 5;   // conv2 weight gradient on the bf16x3 kernel: 144 output tiles x 5 K slices
constexpr int kWg3SplitBf3 = 3;   // conv3: 256 tiles x 3

struct DepthWs {
  unsigned* status;                  // first word of the workspace: the f16x2 overflow guard (common.h), as in the ResNet workspace
  unsigned* bounds;                  // f16x2 scales chosen on the device (F16Scale, nn_kernels.h): bound words 0 w2, 1 w3, 2 dy3, 3 dy2
  float* slots;                      // ... and their {s, 1 / s} slots (2 floats each, same order)
  float *x1, *y1p, *x2, *y2p, *x3, *partial;
  float* x1sel;                      // conv1's raw output at the argmax of every pooling window (sparse backward of layer 1, depth_layer1.hip)
  double* red;
  unsigned char *idx1, *idx2;
  BnBuf bn1, bn2, bn3;
  // backward
  float *dy1, *dy1p, *dy2, *dy2p, *dy3, *dw2o, *wg_ws, *bn_ws, *cs_ws;
  // bf16x3 operands of conv2 (forward and data gradient run on gemm_bf3.hip): planes of pooled1, W2, dY2, flipped W2
  unsigned short *y1p_pl[3], *w2_pl[3], *dy2_pl[3], *w2f_pl[3];
  unsigned short *y2p_pl[3], *w3_pl[3], *dy3_pl[3], *w3f_pl[3];   // conv3 forward / data gradient operands
  float* tail;                              // remainder-tile K-split scratch of the bf16x3 launches
  unsigned short *wg_dyT[3], *wg_pT[3];     // weight-gradient operands (transposed planes), shared by conv2 / conv3
  float* wg_bf3_ws;
  size_t bytes;
};

static BnBuf take_bn(Carver& c, int C) {
  BnBuf b;
  b.scale = c.take<float>(C); b.shift = c.take<float>(C); b.mean = c.take<float>(C); b.invstd = c.take<float>(C);
  return b;
}

static DepthWs depth_carve(void* p, size_t bytes, const DepthGeom& g, bool* ov) {
  Carver c(p, bytes);
  DepthWs w{};
  const long long B = g.B;
  w.status = c.take<unsigned>(64);
  w.bounds = c.take<unsigned>(64);
  w.slots = c.take<float>(64);
  w.x1 = c.take<float>((size_t)g.M1 * 128);
  w.y1p = c.take<float>((size_t)B * g.P1h * g.P1w * 128);
  w.idx1 = c.take<unsigned char>((size_t)B * g.P1h * g.P1w * 128);
  w.x1sel = c.take<float>((size_t)B * g.P1h * g.P1w * 128);
  w.x2 = c.take<float>((size_t)g.M2 * 512);
  w.y2p = c.take<float>((size_t)B * g.P2h * g.P2w * 512);
  w.idx2 = c.take<unsigned char>((size_t)B * g.P2h * g.P2w * 512);
  w.x3 = c.take<float>((size_t)g.M3 * 2048);
  const long long rows1 = std::max<long long>(g.M1 / 64 + 2, 1024 + 2);   // conv1 statistics: one row per workgroup
  size_t part = (size_t)rows1 * 2 * 128;
  part = std::max(part, (size_t)(g.M2 / 64 + 2) * 2 * 512);
  part = std::max(part, (size_t)(g.M3 / 64 + 2) * 2 * 2048);
  w.partial = c.take<float>(part);
  w.red = c.take<double>(std::max(std::max(bn_finalize_ws_doubles((int)rows1, 128), bn_finalize_ws_doubles(g.M2 / 64 + 2, 512)),
                                   bn_finalize_ws_doubles(g.M3 / 64 + 2, 2048)));
  w.bn1 = take_bn(c, 128); w.bn2 = take_bn(c, 512); w.bn3 = take_bn(c, 2048);
  w.dy1 = c.take<float>(std::max((size_t)g.M1 * 128, depth_layer1_sparse_supported(g.c1) ? depth_layer1_sparse_ws_floats(g.c1) : (size_t)0));      // (dense gradient of layer 1, or the scratch of its sparse backward)
  w.dy1p = c.take<float>((size_t)B * g.P1h * g.P1w * 128);
  w.dy2 = c.take<float>((size_t)g.M2 * 512);
  w.dy2p = c.take<float>((size_t)B * g.P2h * g.P2w * 512);
  w.dy3 = c.take<float>((size_t)g.M3 * 2048);
  w.dw2o = c.take<float>((size_t)512 * 1152);
  w.wg_ws = c.take<float>(std::max(std::max((size_t)kWg1Split * 128 * 49, (size_t)kWg2Split * 512 * 1152),
                                   conv1_depth_wgrad_ws_floats(g.c1)));
  w.bn_ws = c.take<float>(bn_backward_ws_floats(2048));
  w.cs_ws = c.take<float>((size_t)256 * 2048);
  for (int i = 0; i < 3; ++i) {
    w.y1p_pl[i] = c.take<unsigned short>((size_t)(B * g.P1h * g.P1w + 1) * 128);
    w.w2_pl[i] = c.take<unsigned short>((size_t)512 * 1152);
    w.dy2_pl[i] = c.take<unsigned short>((size_t)(g.M2 + 1) * 512);
    w.w2f_pl[i] = c.take<unsigned short>((size_t)128 * 4608);
    w.y2p_pl[i] = c.take<unsigned short>((size_t)(g.M3 + 1) * 512);
    w.w3_pl[i] = c.take<unsigned short>((size_t)2048 * 512);
    w.dy3_pl[i] = c.take<unsigned short>((size_t)(g.M3 + 1) * 2048);
    w.w3f_pl[i] = c.take<unsigned short>((size_t)512 * 2048);
    w.wg_dyT[i] = c.take<unsigned short>(std::max(conv_wgrad_bf3_plane_elems(g.c2, 0), conv_wgrad_bf3_plane_elems(g.c3, 0)));
    w.wg_pT[i] = c.take<unsigned short>(std::max(conv_wgrad_bf3_plane_elems(g.c2, 1), conv_wgrad_bf3_plane_elems(g.c3, 1)));
  }
  w.wg_bf3_ws = c.take<float>(std::max(conv_wgrad_bf3_ws_floats(g.c2, kWg2SplitBf3), conv_wgrad_bf3_ws_floats(g.c3, kWg3SplitBf3)));
  w.tail = c.take<float>(std::max((size_t)kResnetTailSlabs * 64 * 64, kGemmTailWsBytes / sizeof(float)));
  w.bytes = c.off;
  if (ov) *ov = c.overflow;
  return w;
}

// ------------------------------------------------------------------------------------------
// ResNet (Bottleneck v1.5) plan
// ------------------------------------------------------------------------------------------
struct RnConv {
  ConvDesc d;
  int layer;     // index into the dic_conv_bn_layer array
};

struct RnPlan {
  std::vector<RnConv> convs;       // execution order == layer order
  size_t max_act = 0, max_partial = 0, max_red = 0;
  int outH = 0, outW = 0;
  int B = 0, H = 0, W = 0;         // input batch / image size
};

static void rn_track(RnPlan& pl, const ConvDesc& d) {
  pl.max_act = std::max(pl.max_act, (size_t)d.M() * d.CO);
  pl.max_partial = std::max(pl.max_partial, (size_t)(d.M() / 32 + 4) * 2 * d.CO);      // (the A-stationary conv3 kernel: partials per 32 rows)
  pl.max_red = std::max(pl.max_red, bn_finalize_ws_doubles(d.M() / 32 + 4, d.CO));
}

static RnPlan resnet_plan(int B, int H, int W, const int* blocks) {
  RnPlan pl;
  pl.B = B; pl.H = H; pl.W = W;
  int li = 0;
  ConvDesc stem{B, H, W, 3, 64, 7, 7, 2, 3, 1};
  pl.convs.push_back({stem, li++});
  rn_track(pl, stem);
  int h = stem.OH(), w = stem.OW();
  h = (h + 2 - 3) / 2 + 1; w = (w + 2 - 3) / 2 + 1;      // maxpool 3x3 s2 p1
  int inpl = 64;
  const int planes_of[4] = {64, 128, 256, 512};
  for (int s = 0; s < 4; ++s) {
    const int planes = planes_of[s];
    for (int b = 0; b < blocks[s]; ++b) {
      const int stride = (s > 0 && b == 0) ? 2 : 1;
      ConvDesc c1{B, h, w, inpl, planes, 1, 1, 1, 0, 0};
      ConvDesc c2{B, h, w, planes, planes, 3, 3, stride, 1, 0};
      const int oh = c2.OH(), ow = c2.OW();
      ConvDesc c3{B, oh, ow, planes, planes * 4, 1, 1, 1, 0, 0};
      pl.convs.push_back({c1, li++}); rn_track(pl, c1);
      pl.convs.push_back({c2, li++}); rn_track(pl, c2);
      pl.convs.push_back({c3, li++}); rn_track(pl, c3);
      if (b == 0) {
        ConvDesc ds{B, h, w, inpl, planes * 4, 1, 1, stride, 0, 0};
        pl.convs.push_back({ds, li++}); rn_track(pl, ds);
      }
      h = oh; w = ow; inpl = planes * 4;
    }
  }
  pl.outH = h; pl.outW = w;
  return pl;
}

struct RnWs {
  unsigned* status;                 // first word of the workspace: the f16x2 overflow guard (common.h; documented in include/dic.h)
  float* act[6];                    // raw conv outputs (conv1, conv2, conv3, downsample) + two fp32 block-input buffers (bf16x3 mode)
  unsigned short* planes[3][3];     // bf16x3 mode: three rotating activation buffers x (hi, mid, lo)
  unsigned short* stem_planes[3];   // bf16x3 mode: zero-padded NHWC4 image planes of the stem (conv_stem_bf3)
  float* partial;
  float* tail;
  double* red;
  BnBuf bn;
  BnBuf bn2, bn3;                   // bf16x3 mode: conv2 / conv3 statistics (conv3's outlive the block: the next block's conv1 applies them)
  BnBuf bn_ds;                      // statistics of the downsample branch (applied inside the block's last BN pass)
  size_t bytes;
};

static RnWs rn_carve(void* p, size_t bytes, const RnPlan& pl, int mode, bool* ov) {
  Carver c(p, bytes);
  RnWs w{};
  w.status = c.take<unsigned>(64);       // (first: dic.h promises the word at offset 0 of the workspace)
  for (int i = 0; i < (mode >= 1 ? 6 : 4); ++i) w.act[i] = c.take<float>(pl.max_act);
  if (mode >= 1)
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) w.planes[i][j] = c.take<unsigned short>(pl.max_act + 2048);   // + one pad row (paired layout)
  if (mode >= 1)
    for (int j = 0; j < 3; ++j) w.stem_planes[j] = c.take<unsigned short>(conv_stem_bf3_plane_elems(pl.B, pl.H, pl.W));
  w.partial = c.take<float>(pl.max_partial);
  w.red = c.take<double>(pl.max_red);
  w.tail = c.take<float>(std::max((size_t)kResnetTailSlabs * 64 * 64, kGemmTailWsBytes / sizeof(float)));
  w.bn = take_bn(c, 2048);
  w.bn2 = take_bn(c, 2048);
  w.bn3 = take_bn(c, 2048);
  w.bn_ds = take_bn(c, 2048);
  w.bytes = c.off;
  if (ov) *ov = c.overflow;
  return w;
}

// conv -> (train: batch statistics from the epilogue partials | eval: running stats) -> scale/shift in bn
static int conv_bn(const float* x, const ConvDesc& d, const dic_conv_bn_layer& L, float* y, float* partial, BnBuf bn,
                   double* red, float* tail, int train_bn, hipStream_t st) {
  int mtiles = 0;
  DIC_TRY(conv_fwd(x, d, L.w, nullptr, y, train_bn ? partial : nullptr, &mtiles, st, 0, tail));
  if (train_bn)
    return bn_finalize_train(partial, mtiles, d.M(), d.CO, L.gamma, L.beta, L.running_mean, L.running_var, bn, red, st);
  return bn_finalize_eval(d.CO, L.gamma, L.beta, L.running_mean, L.running_var, bn, st);
}

#ifdef DIC_EXPERIMENTS
static int g_skip_bn_apply = 0;     // measurement only (codes 124..127): bit 0 = skip the bn_apply_planes launches of c1 / c2 outputs, bit 1 = of block outputs (results
                                    // are then those of stale planes - timing experiment "does the pass ride for free next to other forwards")
void resnet_debug_skip_bn_apply(int on) { g_skip_bn_apply = on; }
#define DIC_BN_APPLY_PLANES(...) do { if (!(g_skip_bn_apply & 1)) DIC_TRY(bn_apply_planes(__VA_ARGS__)); } while (0)       /* c1 / c2 outputs */
#define DIC_BN_APPLY_PLANES_OUT(...) do { if (!(g_skip_bn_apply & 2)) DIC_TRY(bn_apply_planes(__VA_ARGS__)); } while (0)   /* block outputs */
#else
#define DIC_BN_APPLY_PLANES(...) DIC_TRY(bn_apply_planes(__VA_ARGS__))
#define DIC_BN_APPLY_PLANES_OUT(...) DIC_TRY(bn_apply_planes(__VA_ARGS__))
#endif
static int g_strip_stem = 1;        // benchmarking (codes 122/123): 0 = stem on the exact-fp32 gather kernel
static int g_fused_tail_bn = 1;     // benchmarking (codes 120/121): 0 = separate tail fix-up and BN finalize launches
void resnet_debug_fused_tail_bn(int on) { if (on >= 2) g_strip_stem = on - 2; else g_fused_tail_bn = on; }

// conv (bf16x3 planes in, raw fp32 out) -> BN scale/shift
static int conv_bn_bf3(unsigned short* const x_planes[3], const ConvDesc& d, const dic_conv_bn_layer& L, float* y,
                       const RnWs& ws, int train_bn, hipStream_t st, const BnBuf* bn_out = nullptr, int fmt = 0) {
  const BnBuf bn = bn_out ? *bn_out : ws.bn;
  int mtiles = 0;
  const unsigned short* xp[3] = {x_planes[0], x_planes[1], x_planes[2]};
  const unsigned short* wp[3] = {L.w_hi, L.w_mid, L.w_lo};
  int fused = 0;
  const BnFuseArgs fa{L.gamma, L.beta, L.running_mean, L.running_var, bn.scale, bn.shift, bn.mean, bn.invstd,
                      (double)d.M(), kBnEps, kBnMomentum, fmt ? ws.status : nullptr};
  DIC_TRY(conv_fwd_bf3(xp, d, wp, y, train_bn ? ws.partial : nullptr, &mtiles, ws.tail, st, nullptr,
                       (train_bn && g_fused_tail_bn) ? &fa : nullptr, &fused, ACT_NONE, kResnetTailSlabs, fmt,
                       fmt ? 1.0f / (kF16ActScale * L.w_scale) : 1.0f));
  if (train_bn && fused) return DIC_OK;        // statistics were finalized inside the tail fix-up launch
  if (train_bn)
    return bn_finalize_train(ws.partial, mtiles, d.M(), d.CO, L.gamma, L.beta, L.running_mean, L.running_var, bn,
                             ws.red, st, fmt ? ws.status : nullptr);
  return bn_finalize_eval(d.CO, L.gamma, L.beta, L.running_mean, L.running_var, bn, st);
}

// conv1x1 whose input is act(raw * in_bn.scale + in_bn.shift (+ res)) formed inside the kernel's producer waves (conv1x1_fwd_bf3_bn,
// gemm_bf3.hip) -> raw fp32 out + BN scale/shift in `bn`.  Returns 1 when the launch policy keeps the shape off that kernel (nothing
// launched, nothing written).
static int conv_bn_bf3_fused(const float* raw, const BnBuf& in_bn, const float* res, float* act_out, const ConvDesc& d,
                             const dic_conv_bn_layer& L, float* y, const RnWs& ws, int train_bn, hipStream_t st, const BnBuf& bn, int fmt,
                             const BnBuf* res_bn = nullptr) {
  if (!(d.KH == 1 && d.KW == 1 && d.stride == 1 && d.pad == 0 && d.C <= 2048)) return 1;
  int mtiles = 0, fused = 0;
  const unsigned short* wp[3] = {L.w_hi, L.w_mid, L.w_lo};
  const BnFuseArgs fa{L.gamma, L.beta, L.running_mean, L.running_var, bn.scale, bn.shift, bn.mean, bn.invstd,
                      (double)d.M(), kBnEps, kBnMomentum, fmt ? ws.status : nullptr};
  const int rc = conv1x1_fwd_bf3_bn(raw, in_bn.scale, in_bn.shift, res, 1, act_out, d.M(), d.C, wp, d.CO, y, train_bn ? ws.partial : nullptr,
                                    &mtiles, ws.tail, kResnetTailSlabs, st, (train_bn && g_fused_tail_bn) ? &fa : nullptr, &fused, fmt,
                                    fmt ? 1.0f / (kF16ActScale * L.w_scale) : 1.0f, fmt ? ws.status : nullptr,
                                    res_bn ? res_bn->scale : nullptr, res_bn ? res_bn->shift : nullptr);
  if (rc != DIC_OK) return rc;
  if (train_bn && fused) return DIC_OK;
  if (train_bn)
    return bn_finalize_train(ws.partial, mtiles, d.M(), d.CO, L.gamma, L.beta, L.running_mean, L.running_var, bn, ws.red, st,
                             fmt ? ws.status : nullptr);
  return bn_finalize_eval(d.CO, L.gamma, L.beta, L.running_mean, L.running_var, bn, st);
}

#ifdef DIC_EXPERIMENTS
// conv3 (K = 128 / 256, f16x2) on the A-stationary kernel: relu(bn2(raw2)) is formed once per 64-row block inside the kernel (no
// bn_apply_planes pass, no planes).  Returns 1 when the shape is not that kernel's.
static int conv_bn_bf3_astat(const float* raw, const BnBuf& in_bn, const ConvDesc& d, const dic_conv_bn_layer& L, float* y, const RnWs& ws,
                             int train_bn, hipStream_t st, const BnBuf& bn) {
  if (!(d.KH == 1 && d.KW == 1 && d.stride == 1 && d.pad == 0) || !conv1x1_astat_eligible(d.M(), d.C, d.CO)) return 1;
  int mtiles = 0;
  const unsigned short* wp[3] = {L.w_hi, L.w_mid, nullptr};
  DIC_TRY(conv1x1_astat_bn(raw, in_bn.scale, in_bn.shift, 1, d.M(), d.C, wp, d.CO, y, train_bn ? ws.partial : nullptr, &mtiles, st,
                           1.0f / (kF16ActScale * L.w_scale), ws.status));
  if (train_bn)
    return bn_finalize_train(ws.partial, mtiles, d.M(), d.CO, L.gamma, L.beta, L.running_mean, L.running_var, bn, ws.red, st, ws.status);
  return bn_finalize_eval(d.CO, L.gamma, L.beta, L.running_mean, L.running_var, bn, st);
}
#endif

// which BatchNorm-apply passes are folded into the consuming 1x1 convolution (codes 100..103 of dic_debug_force_staged_gemm):
// bit 0 = the block output (relu(bn3(conv3) + identity), consumed by the next block's conv1), bit 1 = conv2's output (consumed by conv3)
static int g_fuse_bn_operand_switch = -1;      // -1 (code 104, default): by operand format - both for bf16x3; block outputs only for f16x2, where the
                                               // matrix-core work per K tile is halved and conv3's eightfold re-transform of its input no longer hides
void resnet_fuse_bn_operand(int mask) { g_fuse_bn_operand_switch = mask < 0 ? -1 : (mask & 3); }
static int g_fuse_res_bn = 1;                  // codes 98 / 99: the downsample branch's BatchNorm applied to the residual inside the on-the-fly 1x1 kernel (f16x2): never / yes (default)
void resnet_fuse_res_bn(int on) { g_fuse_res_bn = on; }
static int g_fuse_bn_halo = 1;                 // codes 108 / 109: conv1's output (consumed by the 3x3 conv2) formed inside the LDS-halo kernel's producer waves: never / where
                                               // that kernel takes the shape (default; f16x2 format, 14x14 maps: 35 of ResNet-152's 50 blocks)
void resnet_fuse_bn_halo(int on) { g_fuse_bn_halo = on; }

// conv2 (3x3, stride 1, 14x14 maps) whose input relu(bn1(raw1)) is formed inside the LDS-halo kernel (conv3x3_fwd_bf3_bn, gemm_bf3.hip) ->
// raw fp32 out + BN scale/shift in `bn`.  Returns 1 when that kernel does not take the shape (nothing launched).
static int conv_bn_bf3_halo_fused(const float* raw, const BnBuf& in_bn, const ConvDesc& d, const dic_conv_bn_layer& L, float* y, const RnWs& ws,
                                  int train_bn, hipStream_t st, const BnBuf& bn, int fmt) {
  if (!fmt || !g_fuse_bn_halo) return 1;
  int mtiles = 0, fused = 0;
  const unsigned short* wp[3] = {L.w_hi, L.w_mid, L.w_lo};
  const BnFuseArgs fa{L.gamma, L.beta, L.running_mean, L.running_var, bn.scale, bn.shift, bn.mean, bn.invstd,
                      (double)d.M(), kBnEps, kBnMomentum, ws.status};
  const int rc = conv3x3_fwd_bf3_bn(raw, in_bn.scale, in_bn.shift, 1, d, wp, y, train_bn ? ws.partial : nullptr, &mtiles, ws.tail, kResnetTailSlabs, st,
                                    (train_bn && g_fused_tail_bn) ? &fa : nullptr, &fused, fmt, 1.0f / (kF16ActScale * L.w_scale), ws.status);
  if (rc != DIC_OK) return rc;
  if (train_bn && fused) return DIC_OK;
  if (train_bn)
    return bn_finalize_train(ws.partial, mtiles, d.M(), d.CO, L.gamma, L.beta, L.running_mean, L.running_var, bn, ws.red, st, ws.status);
  return bn_finalize_eval(d.CO, L.gamma, L.beta, L.running_mean, L.running_var, bn, st);
}

// ResNet forward with the bf16x3 convolution.  A convolution reads either three bf16 planes (written by a bn_apply_planes pass; the 3x3
// and strided layers need that form) or - the stride-1 1x1 layers on the persistent kernel - the raw fp32 output of the layer before
// it, normalised / residual-added / rectified / split on the way into LDS (no pass, no planes in HBM).  Inside a stage that makes
//   conv1 (b >= 1): relu(bn3(raw3 of block b-1) + identity of block b-1) formed on the fly; the same kernel writes that block input
//                   once as fp32 (it is the identity of block b)
//   conv2 (3x3):    planes of relu(bn1(raw1))                    (one pass over M x planes)
//   conv3:          relu(bn2(raw2)) on the fly
// and only the last block of a stage materialises its output as planes (the next stage's strided downsample conv reads them).
static int resnet_fwd_bf3(const dic_conv_bn_layer* layers, const int* blocks, const float* imgs_nchw, int B, int train_bn,
                          float* features, const RnPlan& pl, const RnWs& ws, hipStream_t st, int pool_out, int fmt) {
  const int g_fuse_bn_operand = g_fuse_bn_operand_switch >= 0 ? g_fuse_bn_operand_switch : (fmt ? 1 : 3);
  unsigned* const guard = fmt ? ws.status : nullptr;      // f16x2 overflow guard word (zeroed by resnet_fwd_impl)
  size_t ci = 0;
  float *R2 = ws.act[0], *R3 = ws.act[1], *R1 = ws.act[2], *Cf = ws.act[3];
  float* const IN[2] = {ws.act[4], ws.act[5]};
  float* const X = ws.act[0];             // fp32 copy of the final map for the pooling (R2 is dead by then)
  // fmt 1 (f16x2, gemm_bf3.hip): activation plane sets have two planes, the third pointer is NULL - which is also how the kernels
  // that WRITE planes (bn_apply_planes, bn_relu_maxpool) are told the format.  Those planes are not an exact image of the fp32
  // values, so an identity always travels as fp32 in that mode.
  unsigned short* pset[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) pset[i][j] = (fmt && j == 2) ? nullptr : ws.planes[i][j];
  unsigned short* const* Xp = pset[0];
  unsigned short* const* P1 = pset[1];
  unsigned short* const* P2 = pset[2];
  {   // stem (C_in = 3, 1 % of the FLOPs): exact-fp32 gather kernel, then BN + ReLU + maxpool, then split into planes
    const RnConv& c = pl.convs[ci++];
    const dic_conv_bn_layer& L0 = layers[c.layer];
    const bool stem_f16 = fmt && L0.w_hi && L0.w_mid && !L0.w_lo && L0.w_scale > 0.f;      // layer 0 carries f16x2 strip planes (dic_resnet_pack_stem_weights_f16x2)
    if (g_strip_stem && L0.w_hi && L0.w_mid && (L0.w_lo || stem_f16) && c.d.H % 2 == 0 && c.d.W % 2 == 0) {
      // strip formulation on the bf16x3 kernel (w_hi/mid/lo of layer 0 = dic_resnet_pack_stem_weights)
      int mtiles = 0;
      const unsigned short* wp[3] = {L0.w_hi, L0.w_mid, L0.w_lo};
      DIC_TRY(conv_stem_bf3(imgs_nchw, B, c.d.H, c.d.W, 64, ws.stem_planes, wp, R3, train_bn ? ws.partial : nullptr, &mtiles, st,
                            stem_f16 ? 1 : 0, stem_f16 ? 1.0f / (kF16ActScale * L0.w_scale) : 1.0f, guard));
      if (train_bn)
        DIC_TRY(bn_finalize_train(ws.partial, mtiles, c.d.M(), 64, L0.gamma, L0.beta, L0.running_mean, L0.running_var, ws.bn,
                                  ws.red, st, guard));
      else
        DIC_TRY(bn_finalize_eval(64, L0.gamma, L0.beta, L0.running_mean, L0.running_var, ws.bn, st));
    } else
    DIC_TRY(conv_bn(imgs_nchw, c.d, layers[c.layer], R3, ws.partial, ws.bn, ws.red, ws.tail, train_bn, st));
    // BN + ReLU + maxpool straight into the planes of the first bottleneck's input (no fp32 copy, no split pass)
    DIC_TRY(bn_relu_maxpool(R3, B, c.d.OH(), c.d.OW(), 64, &ws.bn, 1, 3, 2, 1, nullptr, nullptr, st, Xp, guard));
  }
  for (int s = 0; s < 4; ++s) {
    bool pending = false;                  // the block input is not materialised: it is relu(bn3(R3) + pend_res)
    const float* pend_res = nullptr;
    const BnBuf* pend_res_bn = nullptr;    // ... where pend_res is a raw convolution output with this BatchNorm of its own (f16x2: the downsample branch)
    const float* in32_carry = nullptr;     // fp32 copy of the block input written by the previous block's output pass
    for (int b = 0; b < blocks[s]; ++b) {
      const RnConv& c1 = pl.convs[ci++];
      const RnConv& c2 = pl.convs[ci++];
      const RnConv& c3 = pl.convs[ci++];
      const float* in32 = in32_carry;      // the block input as fp32, when it exists (else it exists as exact bf16x3 planes Xp)
      in32_carry = nullptr;
      if (pending) {
        float* in_b = IN[b & 1];
        int rc = conv_bn_bf3_fused(R3, ws.bn3, pend_res, in_b, c1.d, layers[c1.layer], R1, ws, train_bn, st, ws.bn, fmt, pend_res_bn);
        if (rc == 1) {                      // shape not on the persistent kernel: form the input as planes (+ fp32) after all
          DIC_BN_APPLY_PLANES_OUT(R3, pend_res, nullptr, in_b, Xp, c1.d.M(), c1.d.C, ws.bn3, 1, st, pend_res_bn, guard);
          rc = conv_bn_bf3(Xp, c1.d, layers[c1.layer], R1, ws, train_bn, st, nullptr, fmt);
        }
        DIC_TRY(rc);
        in32 = in_b;
      } else {
        DIC_TRY(conv_bn_bf3(Xp, c1.d, layers[c1.layer], R1, ws, train_bn, st, nullptr, fmt));
      }
      {
        int rc = conv_bn_bf3_halo_fused(R1, ws.bn, c2.d, layers[c2.layer], R2, ws, train_bn, st, ws.bn2, fmt);
        if (rc == 1) {                      // not the halo kernel's shape (or mode): planes of relu(bn1(raw1)) first
          DIC_BN_APPLY_PLANES(R1, nullptr, nullptr, nullptr, P1, c1.d.M(), c1.d.CO, ws.bn, 1, st, nullptr, guard);
          rc = conv_bn_bf3(P1, c2.d, layers[c2.layer], R2, ws, train_bn, st, &ws.bn2, fmt);
        }
        DIC_TRY(rc);
      }
      if (b == 0) {
        const RnConv& ds = pl.convs[ci++];
        // downsample branch: raw output + its own statistics; its BatchNorm is applied where the block output is formed
        DIC_TRY(conv_bn_bf3(Xp, ds.d, layers[ds.layer], Cf, ws, train_bn, st, &ws.bn_ds, fmt));
      }
      {
        int rc = 1;
#ifdef DIC_EXPERIMENTS      // parked: the A-stationary conv3 kernel (codes 110 / 111; an explicit folding switch 100..103 keeps the routes it names)
        if (fmt && g_fuse_bn_operand_switch < 0) rc = conv_bn_bf3_astat(R2, ws.bn2, c3.d, layers[c3.layer], R3, ws, train_bn, st, ws.bn3);
#endif
        if (rc == 1 && (g_fuse_bn_operand & 2))
          rc = conv_bn_bf3_fused(R2, ws.bn2, nullptr, nullptr, c3.d, layers[c3.layer], R3, ws, train_bn, st, ws.bn3, fmt);
        if (rc == 1) {
          DIC_BN_APPLY_PLANES(R2, nullptr, nullptr, nullptr, P2, c2.d.M(), c2.d.CO, ws.bn2, 1, st, nullptr, guard);
          rc = conv_bn_bf3(P2, c3.d, layers[c3.layer], R3, ws, train_bn, st, &ws.bn3, fmt);
        }
        DIC_TRY(rc);
      }
      // block output relu(bn3(R3) + identity); identity = the block input (planes Xp, or fp32 when conv1 formed it) or, in the first
      // block of a stage, the downsample branch
      const bool last = (s == 3 && b == blocks[s] - 1);
      // (decided here, by the next conv1's shape: a shape the persistent kernel does not take would pay for the fp32 copy on top of
      //  the planes)
      const bool next_fused = (g_fuse_bn_operand & 1) && b + 1 < blocks[s] && (b == 0 || in32) &&
                              conv1x1_bf3_bn_eligible(c3.d.M(), c3.d.CO, c1.d.CO, kResnetTailSlabs, fmt);
      if (next_fused) {
        // left to the next block's conv1.  The downsample output is normalised in place first (the kernel adds a plain residual)
        // (f16x2: the kernel applies the branch's BatchNorm to the residual itself - switch 99; else an in-place pass over it first)
        pend_res_bn = (b == 0 && fmt && g_fuse_res_bn) ? &ws.bn_ds : nullptr;
        if (b == 0 && !pend_res_bn) DIC_TRY(bn_apply(Cf, nullptr, Cf, c3.d.M(), c3.d.CO, ws.bn_ds, 0, st));
        pend_res = b == 0 ? Cf : in32;
        pending = true;
        continue;
      }
      pending = false;
      const float* identity = b == 0 ? Cf : in32;
      const unsigned short* idp[3] = {Xp[0], Xp[1], Xp[2]};
      // planes into P1 (free again); only the very last block also writes fp32, for the pooling that follows
      // (pool_out == 0: the final map itself is the output, written in place of the fp32 copy) - and, in the f16x2 format, a block
      // whose successor in the stage takes it as identity (those planes are not exact)
      float* y32 = last ? (pool_out == 0 ? features : X) : nullptr;
      if (fmt && b + 1 < blocks[s]) { y32 = IN[(b + 1) & 1]; in32_carry = y32; }
      DIC_BN_APPLY_PLANES_OUT(R3, identity, identity ? nullptr : idp, y32, P1, c3.d.M(), c3.d.CO,
                              ws.bn3, 1, st, b == 0 ? &ws.bn_ds : nullptr, guard);
      std::swap(Xp, P1);
    }
  }
  if (pool_out != 0) DIC_TRY(adaptive_avgpool(X, B, pl.outH, pl.outW, 2048, nullptr, 0, pool_out, features, st));
  // (pool_out == 0: features already hold the [B, outH*outW, 2048] map)
  // the loud end of the overflow guard: downstream ReLUs turn the NaN of an overflowed product into 0, so the raised word - not the
  // values - is what fills the output with NaN (dic.h); one tiny launch, nothing written when the word is clear
  if (guard) DIC_TRY(poison_if_raised(features, (long long)B * (pool_out ? pool_out * pool_out : pl.outH * pl.outW) * 2048, guard, st));
  return DIC_OK;
}

}  // namespace dic

using namespace dic;

extern "C" {

int dic_oihw_to_ohwi(const float* src, float* dst, int O, int I, int KH, int KW, void* stream) {
  DIC_REQUIRE(src && dst && src != dst, "oihw_to_ohwi: bad pointers");
  return oihw_to_ohwi(src, dst, O, I, KH, KW, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
size_t dic_depth_encoder_workspace_bytes(int B, int H, int W) {
  bool ov;
  return depth_carve(nullptr, 0, depth_geom(B, H, W), &ov).bytes;
}

// All bf16x3 weight operands of the depth encoder in one launch, straight from the OIHW parameters (they are trained, so
// this runs every step): conv2 / conv3 forward operands [CO][(kh,kw,c)] and the flipped data-gradient operands
// [C][(KH-1-kh, KW-1-kw, co)], each as paired hi/mid/lo planes.  Replaces seven layout / flip / split launches.
// Output-major: a thread produces 4 consecutive plane elements (8-B stores, coalesced) from 4 gathered fp32 weights
// (the 6.5 MB of parameters stay in L2).
struct DepthWeightPlanes { unsigned short *w2[3], *w2f[3], *w3[3], *w3f[3]; };
__global__ void __launch_bounds__(256) depth_prepare_weights_kernel(const float* __restrict__ w2, const float* __restrict__ w3,
                                                                    DepthWeightPlanes pl, const float* __restrict__ slots) {      // slots != NULL: f16x2 planes, scales slots[0] (w2), slots[2] (w3)
  constexpr int O2 = 512, I2 = 128, O3 = 2048, I3 = 512;
  constexpr long long n2 = (long long)O2 * I2 * 9, n3 = (long long)O3 * I3;
  constexpr long long q2 = n2 / 4, q3 = n3 / 4;              // groups of 4 consecutive elements per plane set
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < 2 * (q2 + q3); e += (long long)gridDim.x * 256) {
    // set 0: w2 forward, 1: w2 flipped, 2: w3 forward, 3: w3 flipped
    int set; long long g = e;
    if (g < q2) set = 0; else if ((g -= q2) < q2) set = 1; else if ((g -= q2) < q3) set = 2; else { g -= q3; set = 3; }
    const int K = set == 0 ? 9 * I2 : set == 1 ? 9 * O2 : set == 2 ? I3 : O3;
    const int kb = K / 32;
    const long long idx = g * 4;                              // linear element index inside the paired plane
    const long long line = idx >> 6;
    const int within = (int)(idx & 63);
    const int row = (int)(line / kb) * 2 + (within >> 5);
    const int k0 = (int)(line % kb) * 32 + (within & 31);
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + u;
      if (set == 0) {            // row = o, k = (kh*3+kw)*I2 + i
        const int tap = k / I2, i = k - tap * I2;
        v[u] = w2[((long long)row * I2 + i) * 9 + tap];
      } else if (set == 1) {     // row = i, k = ((2-kh)*3 + (2-kw))*O2 + o
        const int ftap = k / O2, o = k - ftap * O2;
        v[u] = w2[((long long)o * I2 + row) * 9 + (8 - ftap)];
      } else if (set == 2) {     // row = o, k = i
        v[u] = w3[(long long)row * I3 + k];
      } else {                   // row = i, k = o
        v[u] = w3[(long long)k * I3 + row];
      }
    }
    unsigned short h[4], m[4], l[4];
    if (slots) {
      const float fs = slots[set < 2 ? 0 : 2];
#pragma unroll
      for (int u = 0; u < 4; ++u) split2_f16(v[u], fs, h[u], m[u]);
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) split3_bf16(v[u], h[u], m[u], l[u]);
    }
    unsigned short* const* pp = set == 0 ? pl.w2 : set == 1 ? pl.w2f : set == 2 ? pl.w3 : pl.w3f;
    *reinterpret_cast<uint2*>(pp[0] + idx) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
    *reinterpret_cast<uint2*>(pp[1] + idx) = make_uint2((unsigned)m[0] | ((unsigned)m[1] << 16), (unsigned)m[2] | ((unsigned)m[3] << 16));
    if (!slots) *reinterpret_cast<uint2*>(pp[2] + idx) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
  }
}

static int depth_encoder_fwd_impl(const dic_depth_encoder_weights* w, const dic_depth_bn_state* s, const float* depth, int B,
                                  int H, int W, int train, float* features, void* workspace, size_t workspace_bytes,
                                  void* stream, int pool_out) {
  hipStream_t st = (hipStream_t)stream;
  DIC_REQUIRE(w && s && depth && features && workspace, "depth_encoder_fwd: null pointer");
  DIC_REQUIRE(B > 0 && H >= 43 && W >= 43, "depth_encoder_fwd: input too small (%dx%d)", H, W);
  const DepthGeom g = depth_geom(B, H, W);
  DIC_REQUIRE(g.P2h >= 1 && g.P2w >= 1, "depth_encoder_fwd: input too small");
  bool ov = false;
  DepthWs ws = depth_carve(workspace, workspace_bytes, g, &ov);
  DIC_REQUIRE(!ov, "depth_encoder_fwd: workspace too small (%zu < %zu)", workspace_bytes, ws.bytes);
  int mt = 0;
  // Arithmetic of conv2 / conv3 (forward, data gradient, weight gradient: 91 % of the encoder's FLOPs): the f16x2 operand format of
  // gemm_bf3.hip since round 4 (switch 117, default; 116 = the exact bf16x3 split of rounds 1-3) - half the matrix-core work.  The
  // operands' power-of-two scales: activations kF16ActScale (guarded: status word at offset 0 of this workspace), the TRAINED weights
  // and the gradients a scale chosen on the device every step (F16Scale: the weights' from their exact maximum, a gradient's from a
  // bound its BatchNorm-backward computes before it splits), read by the contraction epilogues from memory - no host synchronisation.
  const int fmt = g_depth_f16x2;
  DIC_TRY(clear_status(ws.status, st));
  const F16Scale s_w2{ws.bounds + 0, ws.slots + 0}, s_w3{ws.bounds + 1, ws.slots + 2};
  if (fmt) {
    DIC_TRY(f16_scale_reset(ws.bounds, 2, st));
    DIC_TRY(f16_scale_from_absmax(w->conv2_w, 512ll * 128 * 9, s_w2, st));
    DIC_TRY(f16_scale_from_absmax(w->conv3_w, 2048ll * 512, s_w3, st));
  }
  {
    DepthWeightPlanes pl;
    for (int i = 0; i < 3; ++i) { pl.w2[i] = ws.w2_pl[i]; pl.w2f[i] = ws.w2f_pl[i]; pl.w3[i] = ws.w3_pl[i]; pl.w3f[i] = ws.w3f_pl[i]; }
    hipLaunchKernelGGL(depth_prepare_weights_kernel, dim3(2048), dim3(256), 0, st, w->conv2_w, w->conv3_w, pl, fmt ? (const float*)ws.slots : nullptr);
    DIC_LAUNCH_CHECK();
  }
  unsigned short* const y1p_out[3] = {ws.y1p_pl[0], ws.y1p_pl[1], fmt ? nullptr : ws.y1p_pl[2]};      // (third pointer NULL = f16x2 planes)
  unsigned short* const y2p_out[3] = {ws.y2p_pl[0], ws.y2p_pl[1], fmt ? nullptr : ws.y2p_pl[2]};
  // conv1 (1->128, k7 s3) + BN + ReLU + maxpool3          (depth_models.py:19-20,36-39)
  if (conv1_depth_supported(g.c1))
    DIC_TRY(conv1_depth_fwd(depth, g.c1, w->conv1_w, w->conv1_b, ws.x1, train ? ws.partial : nullptr, &mt, st));
  else
    DIC_TRY(conv_fwd(depth, g.c1, w->conv1_w, w->conv1_b, ws.x1, train ? ws.partial : nullptr, &mt, st));
  if (train) DIC_TRY(bn_finalize_train(ws.partial, mt, g.M1, 128, w->bn1_w, w->bn1_b, s->rm1, s->rv1, ws.bn1, ws.red, st));
  else DIC_TRY(bn_finalize_eval(128, w->bn1_w, w->bn1_b, s->rm1, s->rv1, ws.bn1, st));
  DIC_TRY(bn_relu_maxpool(ws.x1, B, g.H1, g.W1, 128, &ws.bn1, 1, 3, 3, 0, ws.y1p, ws.idx1, st, y1p_out, ws.status, ws.x1sel));
  // conv2 (128->512, k3) + BN + ReLU + maxpool3            (:21-22,40-43)
  //   on the bf16x3 kernel (fp32-accurate, ~1.4x the exact-fp32 MFMA rate): split the pooled activations and W2
  {
    const unsigned short* xp[3] = {ws.y1p_pl[0], ws.y1p_pl[1], fmt ? nullptr : ws.y1p_pl[2]};
    const unsigned short* wp[3] = {ws.w2_pl[0], ws.w2_pl[1], fmt ? nullptr : ws.w2_pl[2]};
    DIC_TRY(conv_fwd_bf3(xp, g.c2, wp, ws.x2, train ? ws.partial : nullptr, &mt, ws.tail, st, w->conv2_b, nullptr, nullptr,
                         ACT_NONE, kResnetTailSlabs, fmt, 1.0f / kF16ActScale, fmt ? s_w2.slot + 1 : nullptr));
  }
  if (train) DIC_TRY(bn_finalize_train(ws.partial, mt, g.M2, 512, w->bn2_w, w->bn2_b, s->rm2, s->rv2, ws.bn2, ws.red, st));
  else DIC_TRY(bn_finalize_eval(512, w->bn2_w, w->bn2_b, s->rm2, s->rv2, ws.bn2, st));
  DIC_TRY(bn_relu_maxpool(ws.x2, B, g.H2, g.W2, 512, &ws.bn2, 1, 3, 3, 0, ws.y2p, ws.idx2, st, y2p_out, ws.status));
  // conv3 (512->2048, k1) + BN + ReLU + AdaptiveAvgPool(14) -> [B,196,2048]   (:23-24,44-47,54)
  //   on the bf16x3 kernel as well (1x1: OIHW == OHWI, so conv3_w is split as it stands)
  {
    const unsigned short* xp[3] = {ws.y2p_pl[0], ws.y2p_pl[1], fmt ? nullptr : ws.y2p_pl[2]};
    const unsigned short* wp[3] = {ws.w3_pl[0], ws.w3_pl[1], fmt ? nullptr : ws.w3_pl[2]};
    DIC_TRY(conv_fwd_bf3(xp, g.c3, wp, ws.x3, train ? ws.partial : nullptr, &mt, ws.tail, st, w->conv3_b, nullptr, nullptr,
                         ACT_NONE, kResnetTailSlabs, fmt, 1.0f / kF16ActScale, fmt ? s_w3.slot + 1 : nullptr));
  }
  if (train) DIC_TRY(bn_finalize_train(ws.partial, mt, g.M3, 2048, w->bn3_w, w->bn3_b, s->rm3, s->rv3, ws.bn3, ws.red, st));
  else DIC_TRY(bn_finalize_eval(2048, w->bn3_w, w->bn3_b, s->rm3, s->rv3, ws.bn3, st));
  //   pool_out == 0: BN + ReLU only, output = the [B, P2h*P2w, 2048] map (requires a square map)
  DIC_REQUIRE(pool_out != 0 || g.P2h == g.P2w, "depth_encoder_fwd_map: the feature map must be square");
  DIC_TRY(adaptive_avgpool(ws.x3, B, g.P2h, g.P2w, 2048, &ws.bn3, 1, pool_out ? pool_out : g.P2h, features, st));
  if (fmt)      // the loud end of the overflow guard, as in dic_resnet_fwd: NaN features when a pooled activation left the fp16 range
    DIC_TRY(poison_if_raised(features, (long long)B * (pool_out ? pool_out * pool_out : g.P2h * g.P2w) * 2048, ws.status, st));
  return DIC_OK;
}

int dic_depth_encoder_fwd(const dic_depth_encoder_weights* w, const dic_depth_bn_state* s, const float* depth, int B,
                          int H, int W, int train, float* features, void* workspace, size_t workspace_bytes,
                          void* stream) {
  return depth_encoder_fwd_impl(w, s, depth, B, H, W, train, features, workspace, workspace_bytes, stream, 14);
}

int dic_depth_encoder_fwd_map(const dic_depth_encoder_weights* w, const dic_depth_bn_state* s, const float* depth, int B,
                              int H, int W, int train, float* feature_map, void* workspace, size_t workspace_bytes,
                              void* stream) {
  return depth_encoder_fwd_impl(w, s, depth, B, H, W, train, feature_map, workspace, workspace_bytes, stream, 0);
}

static int depth_encoder_bwd_impl(const dic_depth_encoder_weights* w, const float* depth, const float* d_features, int B, int H,
                                  int W, const dic_depth_encoder_grads* gr, void* workspace, size_t workspace_bytes,
                                  void* stream, int pool_out) {
  hipStream_t st = (hipStream_t)stream;
  DIC_REQUIRE(w && depth && d_features && gr && workspace, "depth_encoder_bwd: null pointer");
  const DepthGeom g = depth_geom(B, H, W);
  bool ov = false;
  DepthWs ws = depth_carve(workspace, workspace_bytes, g, &ov);
  DIC_REQUIRE(!ov, "depth_encoder_bwd: workspace too small");
  // (same arithmetic as the forward that filled this workspace: its weight planes and weight scale slots are reused here)
  const int fmt = g_depth_f16x2;
  F16Scale s_dy3{ws.bounds + 2, ws.slots + 4}, s_dy2{ws.bounds + 3, ws.slots + 6};
  const float *inv_w2 = ws.slots + 1, *inv_w3 = ws.slots + 3;
  if (fmt) DIC_TRY(f16_scale_reset(ws.bounds + 2, 2, st));
  unsigned short* const dy3_out[3] = {ws.dy3_pl[0], ws.dy3_pl[1], fmt ? nullptr : ws.dy3_pl[2]};
  unsigned short* const dy2_out[3] = {ws.dy2_pl[0], ws.dy2_pl[1], fmt ? nullptr : ws.dy2_pl[2]};
  // layer 3
  DIC_TRY(adaptive_avgpool_bwd(d_features, B, g.P2h, g.P2w, 2048, pool_out ? pool_out : g.P2h, ws.dy3, st));
  DIC_TRY(relu_mask_bwd(ws.dy3, ws.x3, g.M3, 2048, ws.bn3, st));
  DIC_TRY(bn_backward(ws.dy3, ws.x3, g.M3, 2048, w->bn3_w, ws.bn3, gr->bn3_w, gr->bn3_b, ws.bn_ws, st, dy3_out, fmt ? &s_dy3 : nullptr));
  DIC_TRY(conv_wgrad_bf3(ws.y2p, g.c3, ws.dy3, gr->conv3_w, kWg3SplitBf3, ws.wg_dyT, ws.wg_pT, ws.wg_bf3_ws, st, fmt, s_dy3.slot));   // OHWI == OIHW for 1x1
  DIC_TRY(colsum_rows(ws.dy3, 2048, g.M3, 2048, gr->conv3_b, ws.cs_ws, st));
  // (flipped weight planes w3f_pl / w2f_pl were written by the forward's depth_prepare_weights_kernel: same workspace)
  {
    const unsigned short* dp[3] = {ws.dy3_pl[0], ws.dy3_pl[1], fmt ? nullptr : ws.dy3_pl[2]};
    const unsigned short* wp[3] = {ws.w3f_pl[0], ws.w3f_pl[1], fmt ? nullptr : ws.w3f_pl[2]};
    DIC_TRY(conv_dgrad_s1_bf3(dp, g.c3, wp, ws.dy2p, st, ws.tail, kResnetTailSlabs, fmt, fmt ? s_dy3.slot + 1 : nullptr, fmt ? inv_w3 : nullptr));
  }
  // layer 2
  DIC_TRY(bn_pool_backward(ws.dy2p, ws.idx2, ws.x2, B, g.H2, g.W2, 512, 3, w->bn2_w, ws.bn2, gr->bn2_w, gr->bn2_b,
                           ws.bn_ws, ws.dy2, st, dy2_out, fmt ? &s_dy2 : nullptr));
  DIC_TRY(conv_wgrad_bf3(ws.y1p, g.c2, ws.dy2, ws.dw2o, kWg2SplitBf3, ws.wg_dyT, ws.wg_pT, ws.wg_bf3_ws, st, fmt, s_dy2.slot));
  DIC_TRY(ohwi_to_oihw(ws.dw2o, gr->conv2_w, 512, 128, 3, 3, st));
  DIC_TRY(colsum_rows(ws.dy2, 512, g.M2, 512, gr->conv2_b, ws.cs_ws, st));
  {
    const unsigned short* dp[3] = {ws.dy2_pl[0], ws.dy2_pl[1], fmt ? nullptr : ws.dy2_pl[2]};
    const unsigned short* wp[3] = {ws.w2f_pl[0], ws.w2f_pl[1], fmt ? nullptr : ws.w2f_pl[2]};
    DIC_TRY(conv_dgrad_s1_bf3(dp, g.c2, wp, ws.dy1p, st, ws.tail, kResnetTailSlabs, fmt, fmt ? s_dy2.slot + 1 : nullptr, fmt ? inv_w2 : nullptr));
  }
  // layer 1 (no data gradient: the depth map is detached, depth_train.py:204)
  if (g_l1_sparse && depth_layer1_sparse_supported(g.c1))      // round 4: no full-size gradient, no pass over x1 (depth_layer1.hip; switch 181)
    return depth_layer1_backward_sparse(depth, g.c1, ws.dy1p, ws.idx1, ws.x1sel, w->conv1_w, w->conv1_b, w->bn1_w, ws.bn1, gr->bn1_w, gr->bn1_b,
                                        gr->conv1_w, gr->conv1_b, ws.bn_ws, ws.dy1, ws.cs_ws, st);
  DIC_TRY(bn_pool_backward(ws.dy1p, ws.idx1, ws.x1, B, g.H1, g.W1, 128, 3, w->bn1_w, ws.bn1, gr->bn1_w, gr->bn1_b,
                           ws.bn_ws, ws.dy1, st));
  if (conv1_depth_supported(g.c1)) {
    DIC_TRY(conv1_depth_wgrad(depth, g.c1, ws.dy1, gr->conv1_w, gr->conv1_b, ws.wg_ws, ws.cs_ws, st));   // C_in = 1: OHWI == OIHW
  } else {
    DIC_TRY(conv_wgrad(depth, g.c1, ws.dy1, gr->conv1_w, kWg1Split, ws.wg_ws, st));
    DIC_TRY(colsum_rows(ws.dy1, 128, g.M1, 128, gr->conv1_b, ws.cs_ws, st));
  }
  return DIC_OK;
}

int dic_depth_encoder_bwd(const dic_depth_encoder_weights* w, const float* depth, const float* d_features, int B, int H,
                          int W, const dic_depth_encoder_grads* gr, void* workspace, size_t workspace_bytes,
                          void* stream) {
  return depth_encoder_bwd_impl(w, depth, d_features, B, H, W, gr, workspace, workspace_bytes, stream, 14);
}

int dic_depth_encoder_bwd_map(const dic_depth_encoder_weights* w, const float* depth, const float* d_feature_map, int B,
                              int H, int W, const dic_depth_encoder_grads* gr, void* workspace, size_t workspace_bytes,
                              void* stream) {
  return depth_encoder_bwd_impl(w, depth, d_feature_map, B, H, W, gr, workspace, workspace_bytes, stream, 0);
}

// Diagnostic: copy out the selections of the last forward on this workspace (see include/dic.h).
int dic_depth_encoder_inspect(const void* workspace, size_t workspace_bytes, int B, int H, int W, int which, void* out,
                              long long* n_out, void* stream) {
  DIC_REQUIRE(workspace && B > 0 && which >= 1 && which <= 5, "depth_encoder_inspect: bad arguments");
  const DepthGeom g = depth_geom(B, H, W);
  bool ov = false;
  DepthWs ws = depth_carve(const_cast<void*>(workspace), workspace_bytes, g, &ov);
  DIC_REQUIRE(!ov, "depth_encoder_inspect: workspace too small");
  const long long n = which <= 2 ? (long long)B * g.P1h * g.P1w * 128
                      : which <= 4 ? (long long)B * g.P2h * g.P2w * 512 : g.M3 * 2048;
  if (n_out) *n_out = n;
  if (!out) return DIC_OK;
  if (which == 5) return relu_mask_export(ws.x3, g.M3, 2048, ws.bn3, (unsigned char*)out, (hipStream_t)stream);
  const void* src = which == 1 ? (const void*)ws.y1p : which == 2 ? (const void*)ws.idx1 : which == 3 ? (const void*)ws.y2p
                                                                                                      : (const void*)ws.idx2;
  const size_t bytes = (size_t)n * ((which & 1) ? sizeof(float) : 1);
  DIC_CHECK_HIP(hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return DIC_OK;
}

// ---------------------------------------------------------------------------------------------
int dic_resnet_pack_stem_weights(const float* w_oihw, float* scratch_f32, uint16_t* w_hi, uint16_t* w_mid, uint16_t* w_lo,
                                 void* stream) {
  DIC_REQUIRE(w_oihw && scratch_f32 && w_hi && w_mid && w_lo, "resnet_pack_stem_weights: null pointer");
  unsigned short* wp[3] = {w_hi, w_mid, w_lo};
  return conv_stem_pack_weights(w_oihw, 64, scratch_f32, wp, (hipStream_t)stream);
}

int dic_resnet_pack_stem_weights_f16x2(const float* w_oihw, float* scratch_f32, uint16_t* w_h1, uint16_t* w_h2, float w_scale, void* stream) {
  DIC_REQUIRE(w_oihw && scratch_f32 && w_h1 && w_h2 && w_scale > 0.f, "resnet_pack_stem_weights_f16x2: null pointer / scale");
  unsigned short* wp[3] = {w_h1, w_h2, nullptr};
  return conv_stem_pack_weights(w_oihw, 64, scratch_f32, wp, (hipStream_t)stream, w_scale);
}

int dic_resnet_num_layers(const int* blocks) {
  int n = 1;
  for (int s = 0; s < 4; ++s) n += 3 * blocks[s] + 1;
  return n;
}

size_t dic_resnet_workspace_bytes(int B, int H, int W, const int* blocks, int mode) {
  bool ov;
  return rn_carve(nullptr, 0, resnet_plan(B, H, W, blocks), mode, &ov).bytes;
}

static int resnet_fwd_impl(const dic_conv_bn_layer* layers, int n_layers, const int* blocks, const float* imgs_nchw, int B,
                           int H, int W, int train_bn, int mode, float* features, void* workspace, size_t workspace_bytes,
                           void* stream, int pool_out) {
  hipStream_t st = (hipStream_t)stream;
  DIC_REQUIRE(layers && blocks && imgs_nchw && features && workspace, "resnet_fwd: null pointer");
  DIC_REQUIRE(n_layers == dic_resnet_num_layers(blocks), "resnet_fwd: expected %d conv+bn layers, got %d",
              dic_resnet_num_layers(blocks), n_layers);
  DIC_REQUIRE(B > 0 && H >= 32 && W >= 32, "resnet_fwd: bad input size");
  const RnPlan pl = resnet_plan(B, H, W, blocks);
  bool ov = false;
  DIC_REQUIRE(mode >= 0 && mode <= 2, "resnet_fwd: mode must be 0 (exact-fp32 MFMA), 1 (bf16x3 split MFMA) or 2 (f16x2 split MFMA)");
  RnWs ws = rn_carve(workspace, workspace_bytes, pl, mode, &ov);
  DIC_REQUIRE(!ov, "resnet_fwd: workspace too small (%zu < %zu)", workspace_bytes, ws.bytes);
  DIC_TRY(clear_status(ws.status, st));      // status word (first bytes of the workspace): cleared at the start of every forward
  if (mode >= 1) {
    for (int i = 1; i < n_layers; ++i) {
      DIC_REQUIRE(layers[i].w_hi && layers[i].w_mid, "resnet_fwd: split-operand modes need split weights");
      if (mode == 1) DIC_REQUIRE(layers[i].w_lo, "resnet_fwd: bf16x3 mode needs three weight planes");
      else DIC_REQUIRE(layers[i].w_scale > 0.f, "resnet_fwd: f16x2 mode needs the scale of the weight planes (layer %d)", i);
    }
    return resnet_fwd_bf3(layers, blocks, imgs_nchw, B, train_bn, features, pl, ws, st, pool_out, mode == 2);
  }

  size_t ci = 0;
  float *X = ws.act[0], *A = ws.act[1], *Bf = ws.act[2], *Cf = ws.act[3];
  // stem: conv7x7 s2 + BN + ReLU + maxpool 3x3 s2 p1
  {
    const RnConv& c = pl.convs[ci++];
    DIC_TRY(conv_bn(imgs_nchw, c.d, layers[c.layer], A, ws.partial, ws.bn, ws.red, ws.tail, train_bn, st));
    DIC_TRY(bn_relu_maxpool(A, B, c.d.OH(), c.d.OW(), 64, &ws.bn, 1, 3, 2, 1, X, nullptr, st));
  }
  for (int s = 0; s < 4; ++s)
    for (int b = 0; b < blocks[s]; ++b) {
      const RnConv& c1 = pl.convs[ci++];
      const RnConv& c2 = pl.convs[ci++];
      const RnConv& c3 = pl.convs[ci++];
      DIC_TRY(conv_bn(X, c1.d, layers[c1.layer], A, ws.partial, ws.bn, ws.red, ws.tail, train_bn, st));
      DIC_TRY(bn_apply(A, nullptr, A, c1.d.M(), c1.d.CO, ws.bn, 1, st));
      DIC_TRY(conv_bn(A, c2.d, layers[c2.layer], Bf, ws.partial, ws.bn, ws.red, ws.tail, train_bn, st));
      DIC_TRY(bn_apply(Bf, nullptr, Bf, c2.d.M(), c2.d.CO, ws.bn, 1, st));
      const float* identity = X;
      if (b == 0) {
        const RnConv& ds = pl.convs[ci++];
        DIC_TRY(conv_bn(X, ds.d, layers[ds.layer], Cf, ws.partial, ws.bn, ws.red, ws.tail, train_bn, st));
        DIC_TRY(bn_apply(Cf, nullptr, Cf, ds.d.M(), ds.d.CO, ws.bn, 0, st));
        identity = Cf;
      }
      DIC_TRY(conv_bn(Bf, c3.d, layers[c3.layer], A, ws.partial, ws.bn, ws.red, ws.tail, train_bn, st));
      DIC_TRY(bn_apply(A, identity, Bf, c3.d.M(), c3.d.CO, ws.bn, 1, st));      // out = relu(bn3 + identity)
      std::swap(X, Bf);
    }
  // AdaptiveAvgPool2d(14) + permute(0,2,3,1).flatten(1,2): NHWC already is [B,196,2048]
  if (pool_out == 0) {
    DIC_CHECK_HIP(hipMemcpyAsync(features, X, sizeof(float) * (size_t)B * pl.outH * pl.outW * 2048, hipMemcpyDeviceToDevice, st));
    return DIC_OK;
  }
  DIC_TRY(adaptive_avgpool(X, B, pl.outH, pl.outW, 2048, nullptr, 0, pool_out, features, st));
  return DIC_OK;
}

int dic_resnet_fwd(const dic_conv_bn_layer* layers, int n_layers, const int* blocks, const float* imgs_nchw, int B,
                   int H, int W, int train_bn, int mode, float* features, void* workspace, size_t workspace_bytes,
                   void* stream) {
  return resnet_fwd_impl(layers, n_layers, blocks, imgs_nchw, B, H, W, train_bn, mode, features, workspace,
                         workspace_bytes, stream, 14);
}

int dic_resnet_fwd_map(const dic_conv_bn_layer* layers, int n_layers, const int* blocks, const float* imgs_nchw, int B,
                       int H, int W, int train_bn, int mode, float* feature_map, void* workspace, size_t workspace_bytes,
                       void* stream) {
  return resnet_fwd_impl(layers, n_layers, blocks, imgs_nchw, B, H, W, train_bn, mode, feature_map, workspace,
                         workspace_bytes, stream, 0);
}

}  // extern "C"
