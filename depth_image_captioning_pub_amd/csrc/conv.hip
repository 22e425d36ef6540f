#include "conv.h"

namespace dic {

int conv_mtiles(const ConvDesc& d, int force_tile) {
  const int tile = force_tile ? force_tile : gemm_pick_tile(d.M(), d.CO);
  return ceil_div(d.M(), tile);
}

int conv_fwd(const float* x, const ConvDesc& d, const float* w, const float* bias, float* y, float* bn_partial,
             int* mtiles_out, hipStream_t st, int force_tile, float* tail_ws) {
  DIC_REQUIRE(d.OH() > 0 && d.OW() > 0, "conv: empty output");
  GemmParams p{};
  p.M = d.M(); p.N = d.CO; p.K = d.K();
  const bool fast = (!d.in_nchw) && (d.C % 32 == 0);
  p.A = fast ? op_im2col(x, d.geom()) : op_gather(x, d.geom());
  p.B = op_rowk(w, d.K());
  p.ep = ep_store(y, d.CO, bias, ACT_NONE);
  p.ep.stats = bn_partial;
  p.splitk = 1;
  p.tail_ws = tail_ws;
  const int tile = force_tile ? force_tile : gemm_pick_tile(p.M, p.N);
  if (mtiles_out) *mtiles_out = ceil_div(p.M, tile);
  return gemm_launch(p, st, tile);
}

int conv_wgrad(const float* x, const ConvDesc& d, const float* dy, float* dw, int splitk, float* ws, hipStream_t st) {
  GemmParams p{};
  p.M = d.CO; p.N = d.K(); p.K = d.M();
  p.A = op_colk(dy, d.CO);                       // A(co, m) = dY[m*CO + co]
  const bool fast = (!d.in_nchw) && (d.C % 4 == 0);
  p.B = fast ? op_im2col_colk(x, d.geom()) : op_gather_colk(x, d.geom());
  p.ep = ep_store(dw, d.K());
  p.splitk = splitk; p.ws = ws;
  return gemm_launch(p, st, 64);
}

__global__ void __launch_bounds__(256) flip_weights_kernel(const float* __restrict__ w, float* __restrict__ wf,
                                                            int CO, int KH, int KW, int C) {
  // wf[c][KH-1-kh][KW-1-kw][co] = w[co][kh][kw][c]
  const long long total = (long long)CO * KH * KW * C;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int co = (int)(e % CO);
    long long r = e / CO;
    const int kw2 = (int)(r % KW); r /= KW;
    const int kh2 = (int)(r % KH);
    const int c = (int)(r / KH);
    wf[e] = w[(((long long)co * KH + (KH - 1 - kh2)) * KW + (KW - 1 - kw2)) * C + c];
  }
}

int conv_flip_weights(const float* w, const ConvDesc& d, float* wf, hipStream_t st) {
  const long long total = (long long)d.CO * d.K();
  const int blocks = (int)std::min<long long>((total + 255) / 256, 2048);
  hipLaunchKernelGGL(flip_weights_kernel, dim3(blocks), dim3(256), 0, st, w, wf, d.CO, d.KH, d.KW, d.C);
  DIC_LAUNCH_CHECK();
  return DIC_OK;
}

int conv_dgrad_s1(const float* dy, const ConvDesc& d, const float* w_flip, float* dx, hipStream_t st) {
  DIC_REQUIRE(d.stride == 1, "conv_dgrad_s1: stride must be 1");
  DIC_REQUIRE(d.CO % 32 == 0, "conv_dgrad_s1: CO %% 32");
  ConvGeom g{d.OH(), d.OW(), d.CO, d.H, d.W, d.KH, d.KW, 1, d.KH - 1 - d.pad, 0};
  GemmParams p{};
  p.M = d.B * d.H * d.W; p.N = d.C; p.K = d.KH * d.KW * d.CO;
  p.A = op_im2col(dy, g);
  p.B = op_rowk(w_flip, p.K);
  p.ep = ep_store(dx, d.C);
  p.splitk = 1;
  return gemm_launch(p, st, 0);
}

}  // namespace dic
