/* libdic_hip.so - C ABI of the MI355X-native depth-soft captioning hot path.
 *
 * The reference (Kyo-suke-S/Depth_image_captioning_pub) has NO native / FFI boundary: its hot path
 * sits behind plain torch nn.Module objects.  These entry points are what a
 * ctypes binding of that path binds instead; each one cites the reference code it replaces
 * (paths relative to the reference root).  Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (torch allocates; tensor.data_ptr()),
 *     except `lengths`/`batch_sizes` style small int arrays, which are HOST pointers (the reference
 *     keeps `lengths` as a Python list: depth_models.py:153-154);
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*), no hidden synchronisation;
 *   - return 0 on success, a negative code on failure, never throw; dic_last_error() gives the text;
 *   - the library owns no device memory: scratch comes in as (workspace, workspace_bytes), sized by
 *     the matching *_workspace_bytes() query; one call at a time per workspace.
 *   - all floating point is IEEE fp32 (exact-fp32 MFMA), token ids are int64.
 */
#ifndef DIC_H_
#define DIC_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIC_L 196      /* 14x14 annotation cells     (Captioning_models/config.py:11) */
#define DIC_D 2048     /* dim_encoder                (config.py:14) */
#define DIC_A 128      /* dim_attention              (config.py:12) */
#define DIC_E 128      /* dim_embedding              (config.py:13) */
#define DIC_H 128      /* dim_hidden                 (config.py:15) */

/* ABI version of this header.  dic_version() returns the value the library was built with: a binding compares the two before its
 * first call (the Python loader does, depth_image_captioning_pub_amd/_lib.py).  History: 100 = rounds 1-2; 200 (round 4) = the
 * dic_conv_bn_layer struct of round 3 (w_hi / w_mid / w_lo / w_scale: 72 bytes), dic_vit_attention with (workspace, workspace_bytes),
 * the f16x2 overflow guard (status word at offset 0 of the ResNet workspace, dic_*_guarded, dic_split_f16x2_paired_checked),
 * dic_struct_bytes. */
#define DIC_ABI_VERSION 200
int dic_version(void);
const char* dic_last_error(void);
/* sizeof() of the structs of this header as the library sees them, for bindings that mirror them by hand (ctypes.Structure, cgo, JNI):
 * which 0 dic_conv_bn_layer, 1 dic_decoder_weights, 2 dic_decoder_grads, 3 dic_depth_encoder_weights, 4 dic_depth_encoder_grads,
 * 5 dic_depth_bn_state; 0 for an unknown index. */
size_t dic_struct_bytes(int which);

/* ---- generic exact-fp32 MFMA contraction (building block; replaces the aten::addmm / aten::mm
 *      calls under every nn.Linear of the path, e.g. attention.py:84-87, depth_models.py:167,189,197)
 *      C[M,N] (+)= act(A*B^T + bias);  a_colk/b_colk = 1 means that operand is stored K-major
 *      (element (i,k) at p[k*ld + i]) instead of row-major (p[i*ld + k]). */
int dic_gemm_f32(int M, int N, int K, const float* A, long long lda, int a_colk, const float* B, long long ldb,
                 int b_colk, float* C, long long ldc, const float* bias, int act, int accumulate, int splitk,
                 float* workspace, size_t workspace_bytes, int force_tile, void* stream);
/* act: 0 none, 1 ReLU, 2 sigmoid, 3 GELU (exact erf form = nn.GELU()) */

/* ---- convolution as implicit GEMM, NHWC activations, OHWI weights (replaces aten::conv2d under
 *      Depth_CNN_endoder.features, depth_models.py:19-23,36-47, and torchvision ResNet-152 under
 *      CNNEncoder_Atten.backbone, base_caption_models.py:23-30).  x may be NCHW when in_nchw=1
 *      (first layer: the reference feeds NCHW images).  bn_partial (nullable) receives per-M-tile
 *      column sums / sums of squares [mtiles][2][CO] for train-mode BatchNorm statistics.  tail_ws (nullable):
 *      4 MiB of scratch that lets the launcher K-split the remainder tiles of the last partial round. */
int dic_conv2d_fwd(const float* x, int B, int H, int W, int C, int in_nchw, const float* w_ohwi, const float* bias,
                   int CO, int KH, int KW, int stride, int pad, float* y_nhwc, float* bn_partial, int* mtiles_out,
                   int force_tile, float* tail_ws, void* stream);


/* ---- decoder: CD_RNNDecoderWith{Soft,Hard}Attention (Depth_caption_model/depth_models.py:96-305,
 *      522-789) incl. Soft_Attention / Hard_Attention / Gumbel_softmax (attention.py:6-167).
 *      Pointers follow the reference's state_dict names and native layouts ([out][in] row-major). */
typedef struct dic_decoder_weights {
  const float *enc_att_w, *enc_att_b;     /* attention.encoder_att  [A,D],[A]   (attention.py:64)  */
  const float *dec_att_w, *dec_att_b;     /* attention.decoder_att  [A,H],[A]   (attention.py:67)  */
  const float *full_att_w, *full_att_b;   /* attention.full_att     [1,A],[1]   (attention.py:70)  */
  const float *embed;                     /* embed.weight           [V,E]       (depth_models.py:118) */
  const float *w_ih, *w_hh, *b_ih, *b_hh; /* decode_step            [4H,E+D],[4H,H],[4H],[4H] (:122) */
  const float *init_w, *init_b;           /* init_linear            [2H,D],[2H] (:126) */
  const float *fbeta_w, *fbeta_b;         /* f_beta                 [D,H],[D]   (:129) */
  const float *out_w, *out_b;             /* linear                 [V,H],[V]   (:132) */
} dic_decoder_weights;

typedef struct dic_decoder_grads {        /* same order, written (not accumulated) by dic_decoder_bwd */
  float *enc_att_w, *enc_att_b, *dec_att_w, *dec_att_b, *full_att_w, *full_att_b, *embed;
  float *w_ih, *w_hh, *b_ih, *b_hh, *init_w, *init_b, *fbeta_w, *fbeta_b, *out_w, *out_b;
} dic_decoder_grads;

/* bytes of device scratch ("tape") one forward+backward pair needs; n_packed = sum(dec_lengths) */
size_t dic_decoder_workspace_bytes(int B, int Tmax, int V, int n_packed);

/* forward (depth_models.py:153-207 soft, 580-634 hard-train, 637-689 hard-eval).
 *   feat_rgb/feat_depth: [B,196,2048] contiguous (feat_depth may be NULL = base-* models);
 *   captions: int64 [B,cap_stride] on device; dec_lengths: HOST int[B] = lengths-1, descending;
 *   drop_mult: [B,Tmax,H] multiplier (0 or 1/(1-p)) or NULL for eval (quirk Q6: mask is an input);
 *   mode 0 soft | 1 Gumbel-softmax with temp (attention.py:12-25) | 2 Gumbel-max one-hot (:34-48);
 *   gumbel_u: [Tmax,B,196] uniform draws (modes 1,2);
 *   logits_packed: [n_packed,V] time-major rows (= PackedSequence.data, :204); alphas: [B,Tmax,196]. */
int dic_decoder_fwd(const dic_decoder_weights* w, int V, const float* feat_rgb, const float* feat_depth,
                    const int64_t* captions, int cap_stride, const int* dec_lengths, int B, const float* drop_mult,
                    int mode, const float* gumbel_u, float temp, float* logits_packed, float* alphas, void* workspace,
                    size_t workspace_bytes, void* stream);

/* backward of the call above (autograd of depth_train.py:219 restricted to the decoder): consumes the
 * workspace left by dic_decoder_fwd.  dalphas may be NULL.  d_features [B,196,2048] (nullable) is the
 * gradient w.r.t. BOTH feat_rgb and feat_depth (they are summed, depth_models.py:163). */
int dic_decoder_bwd(const dic_decoder_weights* w, int V, const int64_t* captions, int cap_stride,
                    const int* dec_lengths, int B, const float* drop_mult, int mode, float temp,
                    const float* dlogits_packed, const float* dalphas, const float* alphas,
                    const dic_decoder_grads* g, float* d_features, void* workspace, size_t workspace_bytes,
                    void* stream);

/* Compact 49-cell layout (soft attention only).  At 224x224 both encoders end in a 7x7 map that AdaptiveAvgPool2d(14)
 * replicates 2x2 exactly (base_caption_models.py:27,41; depth_models.py:47,54), so the 196 annotation cells hold 49
 * distinct vectors, softmax_196 = softmax_49 / 4 and ctx = sum_g beta_g F_g.  With cells = 49 feat_rgb / feat_depth /
 * d_features are [B,49,2048] (the 7x7 maps, row-major), every pass over the feature map is 4x smaller, and the results
 * (logits, alphas [B,T,196] in the reference's cell order, all parameter gradients) equal the 196-cell evaluation up to
 * fp32 rounding; d_features is then the gradient w.r.t. the 7x7 map (= the sum over each 2x2 group).
 * cells = 196 is identical to dic_decoder_fwd / dic_decoder_bwd with mode 0.  Same workspace as those. */
int dic_decoder_fwd_cells(const dic_decoder_weights* w, int V, const float* feat_rgb, const float* feat_depth, int cells,
                          const int64_t* captions, int cap_stride, const int* dec_lengths, int B, const float* drop_mult,
                          float* logits_packed, float* alphas, void* workspace, size_t workspace_bytes, void* stream);
int dic_decoder_bwd_cells(const dic_decoder_weights* w, int V, int cells, const int64_t* captions, int cap_stride,
                          const int* dec_lengths, int B, const float* drop_mult, const float* dlogits_packed,
                          const float* dalphas, const float* alphas, const dic_decoder_grads* g, float* d_features,
                          void* workspace, size_t workspace_bytes, void* stream);

/* greedy decoding: batch_sample / sample (depth_models.py:216-305 soft, 698-789 hard with Gumbel-max).
 *   Starts from id_start (<start>), max_length steps, dropout off, token = argmax(linear(h)); the previous
 *   token stays on the device (the reference copies it to the host every step, :298-299).
 *   out_ids: int64 [B,max_length]; alphas_out (nullable when max_length <= 8): [B,max_length,196];
 *   mode 0 soft | 2 hard (gumbel_u [max_length,B,196]). */
size_t dic_decoder_greedy_workspace_bytes(int B, int max_length, int V);
int dic_decoder_greedy(const dic_decoder_weights* w, int V, const float* feat_rgb, const float* feat_depth, int B,
                       long long id_start, int max_length, int mode, const float* gumbel_u, int64_t* out_ids,
                       float* alphas_out, void* workspace, size_t workspace_bytes, void* stream);

/* stand-alone attention module: Soft_Attention.forward (attention.py:81-95), Hard_Attention.forward (:132-148,
 *   mode 1, gumbel_u [B,196], temp) and Hard_Attention.Hard_sample (:150-167, mode 2): feats [B,196,2048],
 *   h [B,128] -> ctx [B,2048], alpha [B,196] (float; one-hot in mode 2). */
size_t dic_attention_workspace_bytes(int B);
int dic_attention_fwd(const float* enc_att_w, const float* enc_att_b, const float* dec_att_w, const float* dec_att_b,
                      const float* full_att_w, const float* full_att_b, const float* feats, const float* h, int B,
                      int mode, const float* gumbel_u, float temp, float* ctx, float* alpha, void* workspace,
                      size_t workspace_bytes, void* stream);

/* autograd backward of the call above (the reference modules are ordinary autograd modules): modes 0 and 1; given
 *   d_ctx [B,2048] and d_alpha [B,196] (nullable) and the forward's alpha, writes the six parameter gradients, d_feats
 *   [B,196,2048] and d_h [B,128]. */
size_t dic_attention_bwd_workspace_bytes(int B);
int dic_attention_bwd(const float* enc_att_w, const float* enc_att_b, const float* dec_att_w, const float* dec_att_b,
                      const float* full_att_w, const float* feats, const float* h, const float* alpha, int B, int mode,
                      float temp, const float* d_ctx, const float* d_alpha, float* g_enc_att_w, float* g_enc_att_b,
                      float* g_dec_att_w, float* g_dec_att_b, float* g_full_att_w, float* g_full_att_b, float* d_feats,
                      float* d_h, void* workspace, size_t workspace_bytes, void* stream);

/* ---- loss of train_Cdepth_soft (depth_train.py:210-216): mean CE over packed tokens
 *      + lam * mean_{B,L}((1 - sum_t alpha)^2).  targets: int64 [n_packed] (device, packed like the
 *      logits).  Writes loss[0] (device), dlogits [n_packed,V] (may alias logits) and, if alphas is
 *      non-NULL, dalphas [B,Tmax,196].  alphas NULL -> CE only (hard path, depth_train.py:530).
 *      ce_grad_scale multiplies dlogits and reg_grad_scale multiplies dalphas (the loss value is unscaled).  Single
 *      device: 1, 1.  Data parallel, rank r of N: n_packed_r / sum_r n_packed_r (token-weighted: the sum over ranks is
 *      the gradient of the global token mean) and 1/N (equal per-rank batch).  A target outside [0, V) - where
 *      F.cross_entropy raises - makes the loss NaN.
 *      scratch: >= (n_packed + B + 8) floats. */
int dic_caption_loss(const float* logits, const int64_t* targets, int n_packed, int V, const float* alphas, int B,
                     int Tmax, float lam, float ce_grad_scale, float reg_grad_scale, float* loss, float* dlogits,
                     float* dalphas, float* scratch, void* stream);
/* packed targets = pack_padded_sequence(captions[:,1:], lengths-1).data (depth_train.py:210-213);
 * `targets` needs room for n_packed int64 + B int32 (the tail holds the device copy of dec_lengths). */
int dic_pack_targets(const int64_t* captions, int cap_stride, const int* dec_lengths, int B, int64_t* targets,
                     void* stream);

/* ---- optimiser: torch.optim.AdamW defaults on a flat fp32 buffer (depth_train.py:136-137,221);
 *      `step` is the 1-based step number. */
int dic_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, int step,
                   float lr, float beta1, float beta2, float eps, float weight_decay, void* stream);
/* The same step, skipped ON THE DEVICE (parameters and both moments untouched) when the device word *skip_if_raised is non-zero
 * (NULL = dic_adamw_step).  Hand it the f16x2 overflow guard word of the forward that produced the gradients (dic_resnet_fwd,
 * below): a step whose features overflowed the fp16 operand planes then leaves the optimiser state as it was, without the host
 * having to look at the word before enqueueing the update. */
int dic_adamw_step_guarded(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, int step,
                           float lr, float beta1, float beta2, float eps, float weight_decay, const uint32_t* skip_if_raised,
                           void* stream);

/* ---- data parallel (the reference itself is single-GPU, config.py:68): the path has ONE exchange step, a
 *      sum all-reduce of the flat gradient buffer between dic_decoder_bwd / dic_depth_encoder_bwd and dic_adamw_step.  The
 *      Python engine issues it through torch.distributed (backend "nccl" = RCCL); these entry points are the same exchange
 *      for callers that bind the C ABI directly.  RCCL is resolved at run time (dlopen; an RCCL the process already holds is
 *      reused).  One process per GPU: rank 0 obtains an id, hands the DIC_COMM_ID_BYTES bytes to the other ranks by any
 *      means (file, socket, MPI), every rank calls dic_comm_create after hipSetDevice.  Gradients must arrive pre-scaled
 *      (dic_caption_loss: grad_scale = this rank's share of the packed tokens, reg_grad_scale = 1 / ranks) so that the SUM is
 *      the gradient of the global loss.  In place, enqueued on `stream`, no host synchronisation. */
#define DIC_COMM_ID_BYTES 128
typedef struct dic_comm dic_comm;
int dic_comm_unique_id(void* id128);
int dic_comm_create(const void* id128, int nranks, int rank, dic_comm** out);
int dic_allreduce_grads(dic_comm* comm, float* flat_grad, long long count, void* stream);
int dic_comm_ranks(const dic_comm* comm, int* nranks, int* rank);
int dic_comm_destroy(dic_comm* comm);
/* Which RCCL the entry points above bound: "rccl=<path of the shared object>;reused ..." when the process already held one (e.g.
 * PyTorch's own librccl.so: that copy is used, so both see the same communicator state) or ";loaded by libdic_hip.so ..." when this
 * library had to dlopen one.  Two different paths in one process (torch's and this one) would mean two RCCL instances: check here.
 * STATUS: the multi-rank path of dic_comm_create / dic_allreduce_grads has only ever executed with ONE rank on hardware (the build
 * pool has 1-GPU boxes); N > 1 is covered over gloo on CPU (tests/test_dp_gloo_cpu.py) and is unverified over RCCL. */
int dic_comm_info(char* buf, size_t bytes);

/* ---- dropout multiplier (nn.Dropout(p) in train mode, depth_models.py:119,197): out[i] = 0 or 1/(1-p),
 *      Philox4x32-10 counter-based stream keyed by (seed, offset). */
int dic_dropout_mask(float* out, long long n, float p, uint64_t seed, uint64_t offset, void* stream);

/* ---- depth encoder: Depth_CNN_endoder (Depth_caption_model/depth_models.py:12-56).
 *      Weights in the reference's native layouts: conv*.weight OIHW, BatchNorm affine + running stats. */
typedef struct dic_depth_encoder_weights {
  const float *conv1_w, *conv1_b, *bn1_w, *bn1_b;   /* [128,1,7,7]   (depth_models.py:19-20) */
  const float *conv2_w, *conv2_b, *bn2_w, *bn2_b;   /* [512,128,3,3] (:21-22) */
  const float *conv3_w, *conv3_b, *bn3_w, *bn3_b;   /* [2048,512,1,1](:23-24) */
} dic_depth_encoder_weights;
typedef struct dic_depth_encoder_grads {
  float *conv1_w, *conv1_b, *bn1_w, *bn1_b, *conv2_w, *conv2_b, *bn2_w, *bn2_b, *conv3_w, *conv3_b, *bn3_w, *bn3_b;
} dic_depth_encoder_grads;
typedef struct dic_depth_bn_state {                 /* running_mean / running_var, updated when train=1 */
  float *rm1, *rv1, *rm2, *rv2, *rm3, *rv3;
} dic_depth_bn_state;

/* Arithmetic (round 4): conv1 in plain fp32 vector FMAs; conv2 / conv3 - forward, data gradient, weight gradient - in the f16x2 operand
 * format of the split-operand kernels (two fp16 planes per operand, three matrix-core products, fp32 accumulation: fp32-level results,
 * tests/test_encoders_gpu.py).  The planes' power-of-two scales: activations 4 (|x| <= 16376 - the first 4 bytes of `workspace` are
 * the same overflow status word as in dic_resnet_fwd, and `features` are filled with NaN when it is raised), the trained weights and
 * the gradients a scale chosen ON THE DEVICE every step (exact maximum / a bound computed by the BatchNorm backward), undone by the
 * contraction epilogues from device memory: no host synchronisation, no range assumption on weights or gradients.
 * dic_debug_force_staged_gemm(116) selects the exact three-way bf16 split of rounds 1-3 instead (117: back to the default). */
size_t dic_depth_encoder_workspace_bytes(int B, int H, int W);
/* forward (depth_models.py:49-56): depth [B,1,H,W] -> features [B,196,2048]; train=1 uses batch
 * statistics and updates the running stats, train=0 uses the running stats. */
int dic_depth_encoder_fwd(const dic_depth_encoder_weights* w, const dic_depth_bn_state* s, const float* depth, int B,
                          int H, int W, int train, float* features, void* workspace, size_t workspace_bytes,
                          void* stream);
/* backward of the train-mode forward above (same workspace): d_features [B,196,2048] -> 12 gradients
 * (written, OIHW like the weights).  No input gradient: the depth map is detached (depth_train.py:204). */
int dic_depth_encoder_bwd(const dic_depth_encoder_weights* w, const float* depth, const float* d_features, int B, int H,
                          int W, const dic_depth_encoder_grads* g, void* workspace, size_t workspace_bytes,
                          void* stream);
/* Depth encoder with the un-replicated output: feature_map [B, P*P, 2048] after BN3 + ReLU (P = 7 at 224x224), and
 * the matching backward taking the gradient w.r.t. that map (dic_decoder_bwd_cells' d_features). */
int dic_depth_encoder_fwd_map(const dic_depth_encoder_weights* w, const dic_depth_bn_state* s, const float* depth, int B,
                              int H, int W, int train, float* feature_map, void* workspace, size_t workspace_bytes,
                              void* stream);
int dic_depth_encoder_bwd_map(const dic_depth_encoder_weights* w, const float* depth, const float* d_feature_map, int B,
                              int H, int W, const dic_depth_encoder_grads* gr, void* workspace, size_t workspace_bytes,
                              void* stream);

/* Diagnostic aid for the parity tests: the selections the last dic_depth_encoder_fwd* call on `workspace` made.  ReLU and
 * max-pool are the only discontinuous operations of the path: an element within fp32 rounding of a tie may be selected
 * differently by two correct fp32 evaluations, which moves gradients by percents; the tests therefore replay the oracle
 * with THESE selections (and check that they differ from the oracle's own only at such ties).
 *   which 1: pooled map 1, float [B,P1h,P1w,128] (NHWC; > 0 <=> the ReLU under the pool passed)   2: its arg-max,
 *   uint8 kh*3+kw inside the 3x3 window;   3 / 4: the same for layer 2, [B,P2h,P2w,512];
 *   5: the layer-3 ReLU decisions as the backward takes them, uint8 [B,P2h,P2w,2048] (1 = passes).
 * *n_out (nullable) receives the element count; out == NULL only queries it.  Device-to-device copy on `stream`. */
int dic_depth_encoder_inspect(const void* workspace, size_t workspace_bytes, int B, int H, int W, int which, void* out,
                              long long* n_out, void* stream);
/* The same for the decoder: the two tensors that decide the attention ReLU of the last dic_decoder_fwd* call on `workspace`
 * (attention.py:84-87; the only other discontinuity of the step):  which 1: P = W_z F + b_z, float [B,cells,128];
 * which 2: q_t = W_h h_t + b_h for every step, float [B,Tmax,128] (rows of finished captions hold stale values).  Unit
 * (b,t,l,a) passes <=> P[b,l,a] + q[b,t,a] > 0 in fp32 - forward and backward kernels both evaluate exactly this sum. */
int dic_decoder_inspect(const void* workspace, size_t workspace_bytes, int B, int Tmax, int V, int n_packed, int cells, int which,
                        float* out, long long* n_out, void* stream);

/* ---- RGB encoder: CNNEncoder_Atten (Base_caption_model/base_caption_models.py:18-45) = torchvision
 *      ResNet-152 (Bottleneck v1.5; blocks = {3,8,36,3}) minus fc, avgpool -> AdaptiveAvgPool2d(14).
 *      Forward only (the reference runs it under @torch.no_grad).  One entry per conv+BN pair in
 *      execution order (stem; per block conv1, conv2, conv3, then downsample for block 0 of a stage);
 *      conv weights in OHWI (use dic_oihw_to_ohwi once on the reference's OIHW tensors). */
typedef struct dic_conv_bn_layer {
  const float* w;                 /* [CO][KH][KW][CI] */
  const float *gamma, *beta;
  float *running_mean, *running_var;
  const uint16_t *w_hi, *w_mid, *w_lo;   /* mode 1: dic_split_bf16x3_paired(w as [CO rows][KH*KW*C]); mode 2: w_hi, w_mid = the two
                                          * planes of dic_split_f16x2_paired(w, scale = w_scale), w_lo NULL (layer 0, the stem, keeps its
                                          * bf16x3 strip planes in both modes); mode 0: NULL */
  float w_scale;                         /* mode 2 only: the power of two the weight planes were scaled by (largest |w| * w_scale in
                                          * (2^13, 2^14]: w_scale = 2^floor(14 - log2 max|w|)) */
} dic_conv_bn_layer;

int dic_oihw_to_ohwi(const float* src, float* dst, int O, int I, int KH, int KW, void* stream);
/* Optional (mode 1): bf16x3 planes of the 7x7 stem weights in the strip order [64][7][8][4] (kw and channel padded with
 * zeros; row-pair interleaved planes of 64 x 224 elements each).  When layer 0 of the table carries them, the stem runs on
 * the bf16x3 kernel over a zero-padded NHWC4 copy of the image instead of the exact-fp32 gather kernel.
 * w_oihw: [64][3][7][7]; scratch_f32: 64*224 floats. */
int dic_resnet_pack_stem_weights(const float* w_oihw, float* scratch_f32, uint16_t* w_hi, uint16_t* w_mid, uint16_t* w_lo,
                                 void* stream);
/* Mode 2 (f16x2): the same strip-ordered stem weights as TWO fp16 planes of w_scale * w (w_scale: a power of two with
 * max|w| * w_scale in (2^13, 2^14], like every mode-2 layer; layer 0 of the table then carries w_hi = w_h1, w_mid = w_h2, w_lo = NULL,
 * w_scale).  The stem then packs the image as two guarded fp16 planes of 4 * x and runs three matrix-core products instead of six. */
int dic_resnet_pack_stem_weights_f16x2(const float* w_oihw, float* scratch_f32, uint16_t* w_h1, uint16_t* w_h2, float w_scale, void* stream);
int dic_resnet_num_layers(const int* blocks);
size_t dic_resnet_workspace_bytes(int B, int H, int W, const int* blocks, int mode);
/* imgs [B,3,H,W] NCHW -> features [B,196,2048].  train_bn=1 reproduces quirk Q1 of the reference
 * (encoder.train() at depth_train.py:161: batch statistics + running-stat updates in the frozen net);
 * train_bn=0 is encoder.eval() (depth_train.py:242).
 * mode 0: exact-fp32 MFMA convolutions; mode 1: fp32-accurate bf16x3 split convolutions (csrc/gemm_bf3.hip; every
 * layer but the C_in=3 stem), same results to fp32 rounding level, ~1.4x the conv throughput at batch 64; mode 2: the same
 * kernels on the f16x2 operand format (two fp16 planes of scaled values, three products; dic_split_f16x2_paired): errors of a few
 * fp32 round-offs per product, inside the envelope of an fp32 evaluation of the network (tests/test_encoders_gpu.py), half the
 * matrix-core work.  Activations are scaled by 4, so a layer input beyond +-16376 does not fit the fp16 planes.
 * OVERFLOW GUARD (mode 2): the first 4 bytes of `workspace` are a status word (uint32).  Every forward clears it first; every kernel
 * that writes f16x2 planes raises it when a value is out of range or not finite, and the BatchNorm statistics kernels raise it on
 * non-finite sums.  When it is raised at the end of the forward, `features` is filled with NaN (the values could otherwise look sane:
 * ReLU turns the NaN of an overflowed product into 0) and the BatchNorm running statistics of the affected channels were left
 * untouched.  A caller reads the word when it next synchronises, or hands its address to dic_adamw_step_guarded /
 * dic_bn_ema_update_guarded so that the step is dropped on the device; the remedy is mode 1 (bf16x3: exact operands, no range limit). */
int dic_resnet_fwd(const dic_conv_bn_layer* layers, int n_layers, const int* blocks, const float* imgs_nchw, int B,
                   int H, int W, int train_bn, int mode, float* features, void* workspace, size_t workspace_bytes,
                   void* stream);
/* Same, but the output is the network's final feature map itself, [B, (H/32)*(W/32), 2048] (7x7 = 49 cells at
 * 224x224), without AdaptiveAvgPool2d(14): the input of the compact decoder layout (dic_decoder_fwd_cells). */
int dic_resnet_fwd_map(const dic_conv_bn_layer* layers, int n_layers, const int* blocks, const float* imgs_nchw, int B,
                       int H, int W, int train_bn, int mode, float* feature_map, void* workspace, size_t workspace_bytes,
                       void* stream);

/* ---- fp32-accurate contraction on the bf16 matrix cores (csrc/gemm_bf3.hip): operands are stored as three bf16
 *      planes hi+mid+lo (exact split of fp32), C = A*B^T from 6 exact bf16 products per k accumulated in fp32. */
int dic_split_bf16x3(const float* x, long long n, uint16_t* hi, uint16_t* mid, uint16_t* lo, void* stream);
/* Row-pair interleaved plane layout (the format the convolutions consume; whole-cache-line LDS-DMA fetches): element
 * (r, k) of a [rows][K] matrix, K % 32 == 0, lives at (((r/2)*(K/32) + k/32)*64 + (r%2)*32 + k%32); each plane holds
 * ((rows+1) & ~1) * K elements (a zero row pads an odd count). */
int dic_split_bf16x3_paired(const float* x, long long rows, int K, uint16_t* hi, uint16_t* mid, uint16_t* lo,
                            void* stream);
/* "f16x2" operand format of the same kernels: two fp16 planes h1 + h2 of scale * x in the same row-pair layout (h1 = rn(scale*x),
 * h2 = rn(scale*x - h1); scale a power of two that puts the largest magnitude of x below 65504 - 2^13 < max <= 2^14 for weights,
 * a fixed 4 for activations) and three matrix-core products instead of six: a few fp32 round-offs per product instead of one, half
 * the matrix-core work.  Used by the frozen ResNet forward (conv mode 2, dic_resnet_fwd). */
int dic_split_f16x2_paired(const float* x, long long rows, int K, float scale, uint16_t* h1, uint16_t* h2, void* stream);
/* ... with the overflow guard: *overflow (device uint32, set to 0 by the caller beforehand) is raised when |scale * x| exceeds 65504
 * anywhere or x is not finite - the planes then hold inf and every product they enter is NaN.  For activations (the DPT front-end
 * checks the word once per forward). */
int dic_split_f16x2_paired_checked(const float* x, long long rows, int K, float scale, uint16_t* h1, uint16_t* h2, uint32_t* overflow,
                                   void* stream);
int dic_gemm_bf16x3_paired(int M, int N, int K, const uint16_t* a_hi, const uint16_t* a_mid, const uint16_t* a_lo,
                           const uint16_t* b_hi, const uint16_t* b_mid, const uint16_t* b_lo, float* C, long long ldc,
                           const float* bias, void* stream);
/* nn.Linear / nn.Conv2d of the DPT front-end on the split-bf16 kernels (fp32-accurate, ~2x the exact-fp32 MFMA rate; dpt.py):
 *   C[M,N] (+)= act(x[M,K] W[N,K]^T + bias)      - the ViT blocks' qkv / proj / fc1 (GELU) / fc2 (vit.py:36-60 -> timm Block),
 *                                                  the 1x1 projections of the reassemble stages (vit.py:345-477);
 *   y[B,OH,OW,CO] = act(conv(x NHWC, w OHWI) + bias) - the residual conv units, fusion-block and head convolutions
 *                                                  (blocks.py:231-341, dpt_depth.py:58-107), C % 32 == 0.
 * Operands are paired planes (dic_split_bf16x3_paired of x viewed as [rows][K] resp. [B*H*W][C], of W as [N][K] resp.
 * [CO][KH*KW*C]); act: 0 none, 1 ReLU, 2 sigmoid, 3 GELU; tail_ws (nullable): kGemmTailWsBytes of scratch as for dic_conv2d_fwd. */
int dic_linear_bf16x3(int M, int N, int K, const uint16_t* const x_planes[3], const uint16_t* const w_planes[3], const float* bias,
                      int act, int accumulate, float* C, long long ldc, void* stream);
int dic_conv2d_bf16x3(const uint16_t* const x_planes[3], int B, int H, int W, int C, const uint16_t* const w_planes[3],
                      const float* bias, int CO, int KH, int KW, int stride, int pad, int act, float* y_nhwc, float* tail_ws,
                      void* stream);
/* ... and on the f16x2 operand format (two planes per operand, dic_split_f16x2_paired; out_scale = 1 / (scale of x * scale of W)):
 * half the matrix-core work, 2^-22 relative representation error per operand */
int dic_linear_f16x2(int M, int N, int K, const uint16_t* const x_planes[2], const uint16_t* const w_planes[2], const float* bias,
                     int act, int accumulate, float* C, long long ldc, float out_scale, void* stream);
int dic_conv2d_f16x2(const uint16_t* const x_planes[2], int B, int H, int W, int Cin, const uint16_t* const w_planes[2],
                     const float* bias, int CO, int KH, int KW, int stride, int pad, int act, float* y_nhwc, float* tail_ws,
                     float out_scale, void* stream);
int dic_gemm_bf16x3(int M, int N, int K, const uint16_t* a_hi, const uint16_t* a_mid, const uint16_t* a_lo,
                    long long lda, const uint16_t* b_hi, const uint16_t* b_mid, const uint16_t* b_lo, long long ldb,
                    float* C, long long ldc, const float* bias, void* stream);

/* ---- DPT-Hybrid depth front-end (BASELINE config 5): DPT_Depthestimator.forward
 *      (Depth_caption_model/DPT_model.py:63-67) = DPTDepthModel(backbone='vitb_rn50_384') (modules/midas/dpt_depth.py:26-107,
 *      blocks.py:231-341, vit.py:36-155,345-477) over timm 0.4.12's vit_base_resnet50_384 (un-vendored: restated).  Frozen,
 *      forward only.  Convolutions / linear layers run on dic_conv2d_fwd / dic_gemm_f32 (act 3 = exact GELU); the entry
 *      points below are the remaining operators.  Activations NHWC ([B,H,W,C]) or [tokens][C], fp32.  The layer sequence is
 *      driven from depth_image_captioning_pub_amd/dpt.py. */
/* StdConv2d(Same).get_weight: out[o,:] = (w[o,:] - mean_o) / (std_o + eps), biased std over the K = C*KH*KW filter taps */
int dic_weight_standardize(const float* w, int O, int K, float eps, float* out, void* stream);
/* F.pad(x, (left, right, top, bottom), value) on NHWC: the asymmetric 'SAME' padding of StdConv2dSame / MaxPool2dSame */
int dic_pad_nhwc(const float* x, int B, int H, int W, int C, int top, int left, int bottom, int right, float value,
                 float* out, void* stream);
/* nn.MaxPool2d(k, stride), no padding (pad first with -inf for 'SAME'), NHWC, C % 4 == 0 */
int dic_maxpool_nhwc(const float* x, int B, int H, int W, int C, int k, int stride, float* out, void* stream);
/* GroupNormAct: out = [relu]( GroupNorm(groups, C, eps)(x) [+ residual] ), x / residual / out [B, HW, C];
 * workspace: dic_groupnorm_workspace_bytes(B, groups) of device scratch (per-slice fp64 partial sums) */
size_t dic_groupnorm_workspace_bytes(int B, int groups);
int dic_groupnorm_nhwc(const float* x, int B, long long HW, int C, int groups, const float* gamma, const float* beta,
                       float eps, const float* residual, int relu, float* out, void* workspace, void* stream);
/* nn.LayerNorm(C, eps) over [rows, C] */
int dic_layernorm(const float* x, long long rows, int C, const float* gamma, const float* beta, float eps, float* out,
                  void* stream);
/* timm Attention core: qkv [B,N,3,heads,64] -> softmax(q k^T / 8) v as [B,N,heads*64].  With `workspace`
 * (dic_vit_attention_workspace_bytes) both products run on the matrix cores in split-bf16 arithmetic (fp32-level accuracy,
 * online softmax in fp32); workspace = NULL keeps the plain fp32 vector kernel. */
size_t dic_vit_attention_workspace_bytes(int B, int N, int heads);
int dic_vit_attention(const float* qkv, int B, int N, int heads, int head_dim, float* out, void* workspace,
                      size_t workspace_bytes, void* stream);
/* F.interpolate(scale_factor=2, mode="bilinear", align_corners=True) on NHWC, C % 4 == 0 (blocks.py:330-334, Interpolate) */
int dic_upsample2x_bilinear_nhwc(const float* x, int B, int H, int W, int C, float* out, void* stream);
/* out[i] = act(a[i] + b[i % period]) (b nullable); act 0 none, 1 ReLU, 3 GELU: residual adds, position embedding, ReLUs */
int dic_add_act(const float* a, const float* b, long long n, long long period, int act, float* out, void* stream);
/* 1x1 convolution to one output channel (+ ReLU): out[r] = [relu](bias + x[r,:] . w), C % 4 == 0 (dpt_depth.py:96-98) */
int dic_pointwise_dot(const float* x, long long rows, int C, const float* w, const float* bias, int relu, float* out,
                      void* stream);

/* ---- host data path on the device: the tensor work of util.collate_func_for_dep
 *      (Captioning_models/util.py:13-17,100-101), DPT_Depthestimator.standardize_depth_map
 *      (Depth_caption_model/DPT_model.py:43-61) and the depth cache lookup (depth_train.py:196-202). */
/* out = (in - mean[c]) / std[c], NCHW, C <= 3; mean3/std3 are HOST arrays (T.Normalize, util.py:13). */
int dic_normalize_images(const float* in, float* out, int B, int C, int H, int W, const float* mean3, const float* std3,
                         void* stream);
/* T.Resize(resize_short, bilinear) + T.CenterCrop(crop) + y*mul+add over `planes` HxW planes (util.py:14-17;
 * align_corners=False, no antialias = torchvision's result when up-scaling 224 -> 384). out: [planes,crop,crop]. */
int dic_resize_bilinear(const float* in, int planes, int H, int W, int resize_short, int crop, float mul, float add,
                        float* out, void* stream);
/* in place: NaN -> 0.5, then per-image (x-min)/(max-min)   (DPT_model.py:50-59); depth: [B, hw]. */
int dic_depth_standardize(float* depth, int B, long long hw, void* stream);
/* running[i] = (1 - momentum) * running[i] + delta[i]: BatchNorm running-statistics update of one batch, applied after a
 * forward that ran ahead with zeroed scratch buffers in the running_mean / running_var slots of its layer table (those then
 * hold momentum * batch statistic); keeps the statistics in batch order with several forwards in flight (engine.py). */
int dic_bn_ema_update(float* running, const float* delta, long long n, float momentum, void* stream);
/* ... skipped on the device when *skip_if_raised != 0 (the f16x2 overflow guard word of the forward that produced `delta`) */
int dic_bn_ema_update_guarded(float* running, const float* delta, long long n, float momentum, const uint32_t* skip_if_raised,
                              void* stream);
/* out[r,:] = table[idx[r],:] (rows of row_floats floats, % 4 == 0; idx int64 on device): depth cache lookup. */
int dic_gather_rows(const float* table, const int64_t* idx, int n, long long row_floats, float* out, void* stream);

/* ---- measurement aid (bench.py roofline): per-launch HIP events around every MFMA contraction launch,
 *      recorded on the launch stream; dic_profile_end synchronises and returns, per kernel instantiation
 *      (key = 1000*(LDS-DMA kernel) + 100*(tile==128) + 10*A_kind + B_kind; 2000 + 10*A_kind = bf16x3 kernel), total milliseconds, algorithmic FLOPs and launches. */
/* Kernel-selection switches (process-global; results stay correct under every code).  libdic_hip.so knows only the codes
 * its own tests use to put two PRODUCT kernels side by side:
 *   11 21 24 20  bf16x3 workgroup tile forced to 64x64 / 128x64 / the persistent warp-specialised 128x128 kernel (wherever its
 *                epilogue applies) / policy default
 *   70 73 79     persistent kernel by policy: never / 1x1 convolutions by CU fill / + gathered convolutions (default)
 *   74 75 78     3x3 convolutions of 14x14 maps on the LDS-halo kernel: always / never / from 128 output tiles (default)
 *   76           accepted, no effect (the persistent kernel's only form here is the warp-specialised one)
 *   80 81        f16x2 1x1 convolutions with the plain epilogue on the 256x128 twelve-wave kernel from 192 such tiles (default) / never
 *   90 91        remainder-round K split of the persistent kernels: off / on (default)
 *   100..104     ResNet forward, BatchNorm-apply passes folded into the consuming 1x1 convolution's operand path: none (every
 *                convolution input is written as planes first) / block outputs only / conv2 outputs only / both / by operand
 *                format (default: both in mode 1, block outputs only in mode 2)
 *   108 109      mode 2: conv1's BatchNorm-apply + ReLU + split formed inside the LDS-halo 3x3 kernel's producer waves (no planes pass): never /
 *                wherever that kernel takes the shape (default)
 *   94 95        3x3 convolutions of 28x28 maps (layer 2) with that on-the-fly operand on the LDS-halo kernel: never (gathered kernel + planes
 *                pass) / by policy (default)
 *   96 97        mode 2: the on-the-fly-operand 1x1 kernel also for 64 output channels (layer 1's conv1; half of its 128-column tile idle):
 *                never (block output written as planes by a pass) / by policy (default)
 *   98 99        mode 2: the downsample branch's own BatchNorm applied to the residual inside the on-the-fly-operand 1x1 kernel (no in-place
 *                pass over that branch): never / yes (default)
 *   92 93        few-tiles launches (every output tile cut into K slices): plain workgroup order / K slice z on XCD z (default)
 *   112 113      mode 2: producer waves of the on-the-fly-operand 1x1 kernel: four / eight (default)
 *   114 115      ... input slots each of its producer waves keeps in flight: four (default) / six
 *   116 117      depth encoder conv2 / conv3 (forward and both gradients): exact bf16x3 split / f16x2 with device-resident scales (default)
 *   182 183      train-mode BatchNorm finalize in two launches (slice sums, then the statistics) above 512 / 1024 (default) rows of partial
 *                sums: ResNet layer 2's 784 rows take the single kernel
 *   180 181      backward of the depth encoder's first layer: three passes over its full-size map and gradient / sparse form without either
 *                (default; csrc/depth_layer1.hip)
 *   120 121      mode 2: computing waves of the 128x128 kernels (LDS-halo 3x3 in its on-the-fly form, persistent 1x1 / gathered) read their
 *                fragments in a block in front of each k-step's matrix instructions / one read in the gap behind each matrix
 *                instruction (default); bit-identical
 *   118 119      weight gradients whose output is 32..255 tiles of 128x128 (the depth encoder's two): 64x64 tiles with the caller's K
 *                split / persistent warp-specialised kernel with every tile cut into K slices (default)
 * Unknown codes are rejected (DIC_ERR_ARG).  Ablation switches and the parked kernels (deep-pipelined / computing-wave-DMA /
 * 256x128 contraction forms, the A-stationary conv3 kernel and the 256x128 kernel's on-the-fly-operand form of round 4, persistent decoder loop, packed-fp32 defect reproducer) are compiled only into the experiments
 * library (python -m depth_image_captioning_pub_amd.build --experiments -> libdic_experiments.so, -DDIC_EXPERIMENTS; codes
 * listed in csrc/api.hip and csrc/gemm_bf3.hip); scripts/ load it with DIC_LIB=experiments, the product never does.
 * bf16x3 key of dic_profile_end: 2000 (f16x2 operand format: 3000) + 10*A_kind (6 = on-the-fly BatchNorm operand) + t, t = 2*(tile_m/64 - 1) + (tile_n/64 - 1) for the plain tiles,
 * 5 = persistent warp-specialised 128x128, 6 = LDS-halo 3x3 (experiments: 4 = deep-pipelined, 7 = 256x128, 8 = computing-wave DMA). */
int dic_debug_force_staged_gemm(int on);
/* Tuning knob (process-global): the persistent split-bf16 convolution kernels (one workgroup per CU, each walking several output
 * tiles) launch at most `max_workgroups` workgroups.  Default 224.  A small value (e.g. 49: every ResNet-152 layer at batch
 * 64 has a multiple of 49 tiles) lets the convolutions of several forwards in flight on different streams occupy disjoint
 * CUs at the same time instead of taking turns on the whole chip; results do not depend on it (same per-tile arithmetic). */
int dic_conv_persistent_grid(int max_workgroups);
int dic_profile_begin(void);
int dic_profile_end(int max_entries, int* keys, double* total_ms, double* total_flops, long long* launches, int* n_out);
/* the same with the algorithmic HBM bytes of the launches per instantiation (operands read once + output written once; the
 * on-the-fly operand: raw input + residual read, fp32 copy written): intensity = flops / bytes against the machine balance decides
 * which roofline bounds the kernel */
int dic_profile_end_bytes(int max_entries, int* keys, double* total_ms, double* total_flops, double* total_bytes, long long* launches,
                          int* n_out);

#ifdef __cplusplus
}
#endif
#endif /* DIC_H_ */
