"""CPU restatement (torch fp32, autograd for gradients) of the depth-soft /
depth-hard Show-Attend-and-Tell hot path.  TEST INFRASTRUCTURE ONLY - see
oracle/__init__.py.  Every function cites the reference lines it follows
(paths relative to /root/reference).

All functions are *functional*: weights come in as a dict keyed by the
reference's ``state_dict`` names, so the same dict can be loaded into the
reference classes (tests/golden/make_golden.py) and into the HIP shims.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
L_CELLS = 196          # 14 x 14 annotation grid  (Captioning_models/config.py:11)
LAMBDA_ALPHA = 0.7     # `lam`  (Depth_caption_model/depth_train.py:25)


# --------------------------------------------------------------------------
# attention  (Captioning_models/attention.py)
# --------------------------------------------------------------------------
def attention_scores(w: Dict[str, Tensor], feats: Tensor, h: Tensor, prefix: str = "attention.", mask: Optional[Tensor] = None,
                     report: Optional[dict] = None) -> Tensor:
    """e[b,l] = w . relu(Wz F[b,l] + bz + Wh h[b] + bh) + b   (attention.py:84-87).
    mask (parity tests only, bool [nb,L,A]): replay THESE ReLU decisions instead of taking them afresh - a unit within
    rounding of the kink may pass in one correct fp32 evaluation and not in another, and each such flip moves the small
    gradients of the attention matrices (and, through dF, of the depth encoder) by the unit's whole term.  `report` receives
    how many decisions differ from this evaluation's own and the largest |pre-activation| among them (the tie evidence)."""
    att1 = F.linear(feats, w[prefix + "encoder_att.weight"], w[prefix + "encoder_att.bias"])
    att2 = F.linear(h, w[prefix + "decoder_att.weight"], w[prefix + "decoder_att.bias"])
    pre = att1 + att2.unsqueeze(1)
    if mask is None:
        act = torch.relu(pre)
    else:
        act = torch.where(mask, pre, torch.zeros_like(pre))
        if report is not None:
            diff = mask != (pre.detach() > 0)
            report["att_relu"] = (report.get("att_relu", (0, 0.0))[0] + int(diff.sum()),
                                  max(report.get("att_relu", (0, 0.0))[1], float((pre.detach().abs() * diff).max()) if bool(diff.any()) else 0.0))
    e = F.linear(act, w[prefix + "full_att.weight"], w[prefix + "full_att.bias"]).squeeze(2)
    return e


def soft_attention(w: Dict[str, Tensor], feats: Tensor, h: Tensor, prefix: str = "attention.", mask: Optional[Tensor] = None,
                   report: Optional[dict] = None) -> Tuple[Tensor, Tensor]:
    """Soft_Attention.forward (attention.py:81-95): alpha = softmax_L(e); ctx = sum_l alpha_l F_l."""
    alpha = attention_scores(w, feats, h, prefix, mask, report).softmax(dim=1)
    ctx = (feats * alpha.unsqueeze(2)).sum(dim=1)
    return ctx, alpha


def gumbel_noise(u: Tensor) -> Tensor:
    """g = -log(-log(u))  (attention.py:18)."""
    return -torch.log(-torch.log(u))


def hard_attention_train(w, feats, h, u: Tensor, temp: Tensor, prefix="attention."):
    """Hard_Attention.forward with the uniform draw `u` made an explicit input
    (attention.py:132-148 + Gumbel_softmax.forward :12-25)."""
    e = attention_scores(w, feats, h, prefix)
    alpha = ((e + gumbel_noise(u)) / temp).softmax(dim=1)
    ctx = (feats * alpha.unsqueeze(2)).sum(dim=1)
    return ctx, alpha


def hard_attention_sample(w, feats, h, u: Tensor, prefix="attention."):
    """Hard_Attention.Hard_sample (attention.py:150-167 + Gumbel_maxtrick :34-48): one-hot alpha (int64)."""
    e = attention_scores(w, feats, h, prefix)
    pos = torch.argmax(e + gumbel_noise(u), dim=1)
    alpha = F.one_hot(pos, num_classes=feats.shape[1])
    ctx = (feats * alpha.unsqueeze(2)).sum(dim=1)
    return ctx, alpha


# --------------------------------------------------------------------------
# decoder  (Depth_caption_model/depth_models.py:96-305, 522-789)
# --------------------------------------------------------------------------
def lstm_cell(w, x, h, c):
    """nn.LSTMCell: gate order i,f,g,o; c' = sig(f) c + sig(i) tanh(g); h' = sig(o) tanh(c')."""
    gates = F.linear(x, w["decode_step.weight_ih"], w["decode_step.bias_ih"]) + \
        F.linear(h, w["decode_step.weight_hh"], w["decode_step.bias_hh"])
    i, f, g, o = gates.chunk(4, dim=1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    return h2, c2


def init_state(w, fused: Tensor):
    """h0,c0 = chunk(init_linear(mean_L F))  (depth_models.py:166-168)."""
    st = F.linear(fused.mean(dim=1), w["init_linear.weight"], w["init_linear.bias"])
    return st.chunk(2, dim=1)


def batch_sizes_of(dec_lengths: Sequence[int]) -> List[int]:
    """bs_valid per step (depth_models.py:182) == PackedSequence.batch_sizes."""
    return [sum(1 for l in dec_lengths if l > t) for t in range(max(dec_lengths))]


def decoder_forward(w: Dict[str, Tensor], feats_rgb: Tensor, feats_depth: Tensor, captions: Tensor,
                    lengths: Sequence[int], drop_mult: Optional[Tensor] = None,
                    hard_u: Optional[Tensor] = None, temp: Optional[Tensor] = None,
                    hard_eval: bool = False, att_masks: Optional[Tensor] = None, report: Optional[dict] = None):
    """CD_RNNDecoderWith{Soft,Hard}Attention.forward / eval_forward
    (depth_models.py:153-207, 580-634, 637-689).

    drop_mult : [B,Tmax,H] multiplier (0 or 1/(1-p)) applied to h before the
                vocabulary projection; None = eval mode (dropout off).
    hard_u    : [Tmax,B,196] uniform draws -> hard (Gumbel) attention; None = soft.
    Returns (packed_logits [N,V] time-major, batch_sizes, alphas [B,Tmax,196]).
    """
    bs = feats_rgb.shape[0]
    emb = F.embedding(captions, w["embed.weight"])                      # :160
    fused = feats_rgb + feats_depth                                     # :163
    h, c = init_state(w, fused)                                          # :166-168
    dec_len = [l - 1 for l in lengths]                                  # :171
    tmax = max(dec_len)
    vocab = w["linear.weight"].shape[0]
    preds = fused.new_zeros((bs, tmax, vocab))
    alphas = fused.new_zeros((bs, tmax, fused.shape[1]))
    for t in range(tmax):                                               # :179
        nb = sum(1 for l in dec_len if l > t)                           # :182
        if hard_u is None:
            ctx, alpha = soft_attention(w, fused[:nb], h[:nb], mask=None if att_masks is None else att_masks[:nb, t], report=report)
        elif hard_eval:
            ctx, alpha = hard_attention_sample(w, fused[:nb], h[:nb], hard_u[t, :nb])
            alpha = alpha.to(fused.dtype)
        else:
            ctx, alpha = hard_attention_train(w, fused[:nb], h[:nb], hard_u[t, :nb], temp)
        gate = torch.sigmoid(F.linear(h[:nb], w["f_beta.weight"], w["f_beta.bias"]))   # :189
        x = torch.cat((emb[:nb, t], gate * ctx), dim=1)                 # :190-192
        h, c = lstm_cell(w, x, h[:nb], c[:nb])                          # :193-194
        hd = h if drop_mult is None else h * drop_mult[:nb, t]          # :197 (dropout on h)
        preds[:nb, t] = F.linear(hd, w["linear.weight"], w["linear.bias"])
        alphas[:nb, t] = alpha                                          # :200-201
    bsz = batch_sizes_of(dec_len)
    packed = torch.cat([preds[:nb, t] for t, nb in enumerate(bsz)], dim=0)  # time-major (:204)
    return packed, bsz, alphas


def pack_targets(captions: Tensor, lengths: Sequence[int]) -> Tensor:
    """pack_padded_sequence(captions[:,1:], lengths-1).data  (depth_train.py:210-213)."""
    dec_len = [l - 1 for l in lengths]
    tg = captions[:, 1:]
    return torch.cat([tg[:nb, t] for t, nb in enumerate(batch_sizes_of(dec_len))], dim=0)


def caption_loss(packed_logits: Tensor, targets: Tensor, alphas: Optional[Tensor], lam: float = LAMBDA_ALPHA):
    """CE(mean over packed tokens) + lam * mean_{B,L}((1 - sum_t alpha)^2)   (depth_train.py:214-216).
    alphas=None -> CE only (hard path, depth_train.py:530-532)."""
    loss = F.cross_entropy(packed_logits, targets)
    if alphas is not None:
        loss = loss + lam * ((1.0 - alphas.sum(dim=1)) ** 2).mean()
    return loss


def batch_sample(w, feats_rgb, feats_depth, id_start: int, max_length: int = 30,
                 hard_u: Optional[Tensor] = None) -> Tensor:
    """Greedy decode, CD_RNNDecoderWithSoftAttention.batch_sample (depth_models.py:259-305);
    hard variant (:742-789) when hard_u [max_length,B,196] is given. Returns int64 [B,max_length]."""
    fused = feats_rgb + feats_depth
    bs = fused.shape[0]
    h, c = init_state(w, fused)
    prev = torch.full((bs,), id_start, dtype=torch.int64)
    out = torch.zeros((bs, max_length), dtype=torch.int64)
    for step in range(max_length):
        e = F.embedding(prev, w["embed.weight"])
        if hard_u is None:
            ctx, _ = soft_attention(w, fused, h)
        else:
            ctx, _ = hard_attention_sample(w, fused, h, hard_u[step])
        gate = torch.sigmoid(F.linear(h, w["f_beta.weight"], w["f_beta.bias"]))
        h, c = lstm_cell(w, torch.cat((e, gate * ctx), dim=1), h, c)
        pred = F.linear(h, w["linear.weight"], w["linear.bias"]).softmax(dim=1)   # :295-296
        prev = pred.argmax(dim=1)
        out[:, step] = prev
    return out


# --------------------------------------------------------------------------
# depth CNN encoder  (Depth_caption_model/depth_models.py:12-56)
# --------------------------------------------------------------------------
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def batch_norm(x, w, prefix, train: bool, state: Optional[Dict[str, Tensor]] = None):
    """nn.BatchNorm2d: batch statistics (biased var) in train mode + running-stat update
    (unbiased var, momentum 0.1); running stats in eval mode."""
    rm, rv = state[prefix + "running_mean"], state[prefix + "running_var"]
    return F.batch_norm(x, rm, rv, w[prefix + "weight"], w[prefix + "bias"], training=train,
                        momentum=BN_MOMENTUM, eps=BN_EPS)


def depth_encoder_forward(w: Dict[str, Tensor], state: Dict[str, Tensor], depth: Tensor, train: bool) -> Tensor:
    """Depth_CNN_endoder.forward (depth_models.py:49-56): conv7s3+BN+ReLU+maxpool3 ->
    conv3+BN+ReLU+maxpool3 -> conv1+BN+ReLU -> AdaptiveAvgPool(14) -> [B,196,2048].
    `state` holds bn{1,2,3}.running_{mean,var}; updated in place when train."""
    x = F.conv2d(depth, w["conv1.weight"], w["conv1.bias"], stride=3)
    x = F.max_pool2d(torch.relu(batch_norm(x, w, "bn1.", train, state)), 3)
    x = F.conv2d(x, w["conv2.weight"], w["conv2.bias"])
    x = F.max_pool2d(torch.relu(batch_norm(x, w, "bn2.", train, state)), 3)
    x = F.conv2d(x, w["conv3.weight"], w["conv3.bias"])
    x = torch.relu(batch_norm(x, w, "bn3.", train, state))
    x = F.adaptive_avg_pool2d(x, 14)
    return x.permute(0, 2, 3, 1).flatten(1, 2)


def _windows3(z: Tensor) -> Tensor:
    """[B,C,H,W] -> the non-overlapping 3x3 windows of max_pool2d(z, 3) as [B,C,PH,PW,9] (index kh*3+kw)."""
    B, C, H, W = z.shape
    ph, pw = H // 3, W // 3
    return z[:, :, :ph * 3, :pw * 3].reshape(B, C, ph, 3, pw, 3).permute(0, 1, 2, 4, 3, 5).reshape(B, C, ph, pw, 9)


def depth_encoder_forward_replay(w: Dict[str, Tensor], state: Dict[str, Tensor], depth: Tensor,
                                 decisions: Dict[str, Tensor]):
    """Depth_CNN_endoder.forward in train mode (depth_models.py:49-56) with the SELECTIONS of its discontinuous
    operations given instead of recomputed: `decisions` (from another evaluation of the same network, e.g. the HIP
    path's dic_depth_encoder_inspect) holds, NHWC-flattened as [B, PH*PW, C]:
      argmax1/argmax2 (uint8 kh*3+kw: which window element each max-pool output takes),
      pooled1/pooled2 (float; > 0 <=> the ReLU under the pool passed), relu3 (uint8: the last ReLU passes).
    ReLU and max-pool are piecewise linear, so with the selections fixed the network is smooth and two fp32 (or fp64)
    evaluations agree to rounding level; near a tie the selections themselves can legitimately differ.  The report says
    how far every given selection is from this evaluation's own choice:
      report[name] = (number of selections differing from this evaluation's own,
                      worst shortfall = max over them of (own best value - selected value) / max|z|  for arg-max,
                                                        |value at the ReLU| / max|z|                   for ReLU).
    A shortfall at fp32 rounding level (<~ 1e-5) means the differing selection is a tie-break, not an error.
    Works in the dtype of `w` / `depth` (fp64 for the yardstick)."""
    report = {}

    def stage(x, i, pooled_key, arg_key):
        z = batch_norm(x, w, f"bn{i}.", True, state)
        win = _windows3(z)                                                     # [B,C,PH,PW,9]
        B, C, ph, pw, _ = win.shape
        idx = decisions[arg_key].reshape(B, ph, pw, C).permute(0, 3, 1, 2).long()
        passed = (decisions[pooled_key].reshape(B, ph, pw, C).permute(0, 3, 1, 2) > 0)
        sel = win.gather(4, idx.unsqueeze(-1)).squeeze(-1)
        scale = float(z.detach().abs().max())
        with torch.no_grad():
            best = win.max(dim=4).values
            own_pass = best > 0
            # the arg-max only matters where the ReLU passes (a clipped window has zero gradient whatever is picked)
            differs = (sel < best) & (passed | own_pass)
            short = float(((best - sel) * differs).max()) / scale if bool(differs.any()) else 0.0
            rdiff = passed != own_pass
            rshort = float((best.abs() * rdiff).max()) / scale if bool(rdiff.any()) else 0.0
            report[f"pool{i}"] = (int(differs.sum()), short)
            report[f"relu{i}"] = (int(rdiff.sum()), rshort)
        return sel * passed.to(sel.dtype)

    x = F.conv2d(depth, w["conv1.weight"], w["conv1.bias"], stride=3)
    x = stage(x, 1, "pooled1", "argmax1")
    x = F.conv2d(x, w["conv2.weight"], w["conv2.bias"])
    x = stage(x, 2, "pooled2", "argmax2")
    x = F.conv2d(x, w["conv3.weight"], w["conv3.bias"])
    z = batch_norm(x, w, "bn3.", True, state)
    B, C, ph, pw = z.shape
    passed = decisions["relu3"].reshape(B, ph, pw, C).permute(0, 3, 1, 2) > 0
    with torch.no_grad():
        rdiff = passed != (z > 0)
        report["relu3"] = (int(rdiff.sum()), float((z.abs() * rdiff).max()) / float(z.abs().max()) if bool(rdiff.any()) else 0.0)
    x = z * passed.to(z.dtype)
    x = F.adaptive_avg_pool2d(x, 14)
    return x.permute(0, 2, 3, 1).flatten(1, 2), report


# --------------------------------------------------------------------------
# RGB encoder: torchvision ResNet-152 (Bottleneck v1.5, layers [3,8,36,3], stride on the 3x3)
# followed by AdaptiveAvgPool2d(14)  (Base_caption_model/base_caption_models.py:18-45).
# PARITY UNPINNED: torchvision is absent from the build container; restated from the public
# architecture definition and checked only against torch's own conv/batch_norm CPU ops.
# --------------------------------------------------------------------------
RESNET152_LAYERS = (3, 8, 36, 3)
RESNET_PLANES = (64, 128, 256, 512)


def resnet_bn(x, w, prefix, train):
    return F.batch_norm(x, w[prefix + "running_mean"], w[prefix + "running_var"],
                        w[prefix + "weight"], w[prefix + "bias"], training=train,
                        momentum=BN_MOMENTUM, eps=BN_EPS)


def resnet152_features(w: Dict[str, Tensor], imgs: Tensor, train_bn: bool,
                       layers: Sequence[int] = RESNET152_LAYERS) -> Tensor:
    """CNNEncoder_Atten.forward (base_caption_models.py:36-45).  `w` uses the Sequential keys of
    `self.backbone` (children()[:-1] of torchvision resnet152): 0=conv1, 1=bn1, 4..7=layer1..4.
    train_bn=True reproduces quirk Q1 (encoder.train() under no_grad: batch statistics and
    running-stat updates in the frozen network; depth_train.py:161)."""
    with torch.no_grad():
        x = F.conv2d(imgs, w["backbone.0.weight"], None, stride=2, padding=3)
        x = torch.relu(resnet_bn(x, w, "backbone.1.", train_bn))
        x = F.max_pool2d(x, 3, stride=2, padding=1)
        for li, nblocks in enumerate(layers):
            for bi in range(nblocks):
                p = f"backbone.{4 + li}.{bi}."
                stride = 2 if (li > 0 and bi == 0) else 1
                idt = x
                y = F.conv2d(x, w[p + "conv1.weight"])
                y = torch.relu(resnet_bn(y, w, p + "bn1.", train_bn))
                y = F.conv2d(y, w[p + "conv2.weight"], stride=stride, padding=1)
                y = torch.relu(resnet_bn(y, w, p + "bn2.", train_bn))
                y = F.conv2d(y, w[p + "conv3.weight"])
                y = resnet_bn(y, w, p + "bn3.", train_bn)
                if bi == 0:
                    idt = F.conv2d(x, w[p + "downsample.0.weight"], stride=stride)
                    idt = resnet_bn(idt, w, p + "downsample.1.", train_bn)
                x = torch.relu(y + idt)
        x = F.adaptive_avg_pool2d(x, 14)                      # 7x7 -> 14x14 replication (Q3)
        return x.permute(0, 2, 3, 1).flatten(1, 2)


# --------------------------------------------------------------------------
# optimiser  (torch.optim.AdamW defaults, depth_train.py:136-137; scheduler never stepped - Q2)
# --------------------------------------------------------------------------
def adamw_step(params: Dict[str, Tensor], grads: Dict[str, Tensor], exp_avg: Dict[str, Tensor],
               exp_avg_sq: Dict[str, Tensor], step: int, lr: float = 1e-3, beta1: float = 0.9,
               beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 0.01) -> None:
    """In-place decoupled-weight-decay Adam, bias-corrected; `step` is the 1-based step count."""
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    for k, p in params.items():
        g = grads[k]
        p.mul_(1.0 - lr * weight_decay)
        exp_avg[k].mul_(beta1).add_(g, alpha=1.0 - beta1)
        exp_avg_sq[k].mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
        denom = (exp_avg_sq[k].sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(exp_avg[k], denom, value=-(lr / bc1))


# --------------------------------------------------------------------------
# one full training step of train_Cdepth_soft's inner loop (depth_train.py:168-221),
# minus the frozen DPT front-end (depth map is an input) - used as the CPU baseline and
# as the step-level parity oracle.
# --------------------------------------------------------------------------
def train_step_soft(dec_w: Dict[str, Tensor], enc_w: Dict[str, Tensor], enc_state: Dict[str, Tensor],
                    feats_rgb: Tensor, depth_map: Tensor, captions: Tensor, lengths: Sequence[int],
                    drop_mult: Optional[Tensor], decisions: Optional[Dict[str, Tensor]] = None,
                    report: Optional[dict] = None, hard_u: Optional[Tensor] = None, temp: Optional[Tensor] = None,
                    att_masks: Optional[Tensor] = None):
    """Forward + loss + backward.  Returns (loss, packed_logits, alphas, grads_dec, grads_enc).
    att_masks (bool [B,Tmax,196,A], soft attention): replay the attention ReLU decisions (attention_scores); `report`
    receives the tie evidence under "att_relu".
    hard_u [Tmax,B,196] + temp: the depth-HARD step of train_Cdepth_hard (depth_train.py:500-560): Gumbel-softmax
    attention with the uniform draws as explicit input, loss = cross-entropy only (:530-532).
    decisions: replay the depth encoder's ReLU / max-pool selections (depth_encoder_forward_replay; `report` receives
    its tie report) instead of taking them afresh - parity tests only."""
    dw = {k: v.detach().clone().requires_grad_(True) for k, v in dec_w.items()}
    ew = {k: v.detach().clone().requires_grad_(True) for k, v in enc_w.items()}
    if decisions is None:
        fd = depth_encoder_forward(ew, enc_state, depth_map.detach(), train=True)     # :204-206
    else:
        fd, rep = depth_encoder_forward_replay(ew, enc_state, depth_map.detach(), decisions)
        if report is not None:
            report.update(rep)
    packed, bsz, alphas = decoder_forward(dw, feats_rgb, fd, captions, lengths, drop_mult, hard_u=hard_u, temp=temp,
                                          att_masks=att_masks, report=report)
    loss = caption_loss(packed, pack_targets(captions, lengths), None if hard_u is not None else alphas)   # :210-216 / :530
    loss.backward()                                                                # :219
    gd = {k: v.grad for k, v in dw.items()}
    ge = {k: v.grad for k, v in ew.items()}
    return loss.detach(), packed.detach(), alphas.detach(), gd, ge


def train_step_base_soft(dec_w: Dict[str, Tensor], feats_rgb: Tensor, captions: Tensor, lengths: Sequence[int],
                         drop_mult: Optional[Tensor], att_masks: Optional[Tensor] = None, report: Optional[dict] = None):
    """One iteration of train_base_soft (Base_caption_model/base_train.py:149-167): RNNDecoderWithSoftAttention forward
    (base_caption_models.py:105-185), CE + 0.7 x regulariser, backward; gradients for the decoder only (the optimiser holds
    decoder.parameters(), base_train.py:115).  The base decoder is the CD_ decoder without the `features + depth_features`
    line (depth_models.py:163 vs base_caption_models.py:132): zero depth features reproduce it exactly (x + 0 = x), which
    is how SURVEY.md 8c defines the oracle for the un-importable base classes.  Returns (loss, packed_logits, alphas, grads)."""
    dw = {k: v.detach().clone().requires_grad_(True) for k, v in dec_w.items()}
    packed, _, alphas = decoder_forward(dw, feats_rgb, torch.zeros_like(feats_rgb), captions, lengths, drop_mult,
                                        att_masks=att_masks, report=report)
    loss = caption_loss(packed, pack_targets(captions, lengths), alphas)               # base_train.py:160-162
    loss.backward()                                                                    # :165
    return loss.detach(), packed.detach(), alphas.detach(), {k: v.grad for k, v in dw.items()}


@torch.no_grad()
def step_logits(dec_w: Dict[str, Tensor], enc_w: Dict[str, Tensor], enc_state: Dict[str, Tensor], feats_rgb: Tensor,
                depth_map: Tensor, captions: Tensor, lengths: Sequence[int], drop_mult: Optional[Tensor]):
    """Forward half of train_step_soft (depth_train.py:204-216) without autograd: (loss, packed_logits).  The parity tests
    evaluate it in fp64 (weights, features, depth map and dropout multiplier as double) as the yardstick of what an fp32
    evaluation of the step can decide."""
    fd = depth_encoder_forward(enc_w, {k: v.clone() for k, v in enc_state.items()}, depth_map, train=True)
    packed, _, alphas = decoder_forward(dec_w, feats_rgb, fd, captions, lengths, drop_mult)
    return caption_loss(packed, pack_targets(captions, lengths), alphas), packed


def rows_undecidable_by_oracle(packed32: Tensor, packed64: Tensor, factor: float = 2.0) -> Tensor:
    """bool [N]: packed rows whose token-id argmax the fp32 ORACLE ITSELF cannot decide - its fp32 and fp64 evaluations pick
    different tokens, or its fp32 top-2 margin is within `factor` x its own max |logit32 - logit64| on that row.  Defined by the
    oracle's two precisions only: nothing of the implementation under test enters.  Everywhere else a
    correct fp32 implementation must reproduce the oracle's argmax exactly."""
    p32, p64 = packed32.double(), packed64.double()
    top2 = p32.topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    own_err = (p32 - p64).abs().max(dim=1).values
    return (p32.argmax(1) != p64.argmax(1)) | (margin <= factor * own_err)
