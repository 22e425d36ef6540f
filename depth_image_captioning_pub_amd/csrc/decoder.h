// Show-Attend-and-Tell decoder (soft + Gumbel "hard" attention): forward, BPTT backward, greedy decode.
// Host orchestration + kernels live in decoder.hip; the C ABI wrappers are at the bottom of that file.
#pragma once
#include "dic.h"
#include "gemm.h"

namespace dic {

constexpr int kL = DIC_L, kD = DIC_D, kA = DIC_A, kE = DIC_E, kH = DIC_H;
constexpr int kG = 4 * kH;              // LSTM gate rows (i,f,g,o)
constexpr int kXK = kE + kD + kH;       // K of the fused LSTM GEMM: [embedding | gate*ctx | h_prev]
constexpr int kNCH = kD / 256;          // D chunks of 256 channels per workgroup (attention kernels)
constexpr int kLCH = 4;                 // L chunks of 49 cells (score-backward kernel)
constexpr int kLc = 49;                 // distinct cells when the 14x14 grid is a 2x2 replication of a 7x7 map (Q3)
constexpr int kS_LSTM = 18;             // split-K of the per-step LSTM gate GEMM (72 K tiles -> 4 per workgroup)
constexpr int kS_DX = 4;                // split-K of the per-step dX GEMM (16 K tiles)

// workspace ("tape") shared by forward and backward of one decoder call
struct DecoderWs {
  // forward / saved for backward
  float *F, *P, *mean, *Wcat, *WcatT, *bcat, *WhT, *WbT, *WzT, *Xall, *Hall, *Call, *Gact, *Qall, *ctx, *gate, *Hdrop;
  float *slab_g, *gemm_ws;
  // backward
  float *dHd, *dG, *slab_dx, *dctx, *dgpre, *dq, *dalp, *pbeta, *dqp, *dwf_acc, *dbf_acc, *dPacc, *carry_dc;
  float *dinit, *dmean, *colsum_ws;
  float *alpha_c, *dalpha_c;     // compact (49-cell) mode: group softmax [B,T,49] and its incoming gradient
  float* dXe;                    // [B,T,E] gradient of the embedded input rows (reduced per token after BPTT)
  int* dlen;
  // persistent forward loop (experiments/decoder_persist.hip; carved in the experiments build only)
  float* Gemb;                   // [B*T, 4H] embedding part of the gate pre-activations + bias
  float* pslab;                  // [2][16][16][4][4H] partial gate pre-activations exchanged per step
  unsigned int* psync;           // [16] arrival counters + [1] status word
  int* poff;                     // [T+1] packed row offsets (device copy of the host plan)
  float* logits_step;
  long long* ids;
  size_t gemm_ws_floats;
  size_t bytes;
};

DecoderWs decoder_carve(void* ws, size_t ws_bytes, int B, int T, int V, int N, bool* overflow);

// Experiments build only (-DDIC_EXPERIMENTS): one launch for all T forward steps (soft attention, B <= 64); see experiments/decoder_persist.hip.  Expects F, P, h0/c0 (slot 0 of
// Hall/Call), the embedding columns of Xall, WhT / WbT / WcatT, Gemb and the device copy of the lengths in the workspace.
bool decoder_persist_eligible(int B, int T, int mode);
int decoder_fwd_persistent(const DecoderWs& ws, const dic_decoder_weights* w, int B, int T, int cells, const float* drop_mult,
                           float* alphas, const int* host_packed_off, hipStream_t st);
void decoder_debug_persistent(int on);     // benchmarking switch: 0 = per-step launches (default), 1 = persistent loop

}  // namespace dic
