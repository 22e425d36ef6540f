"""Thin Python functions over the C ABI (include/dic.h).  torch is used only for device memory and
streams; every number is produced by libdic_hip.so.  Nothing here falls back to torch ops."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

L_CELLS, D_ENC, D_ATT, D_EMB, D_HID = 196, 2048, 128, 128, 128
# Arithmetic of the frozen ResNet-152's convolutions everywhere a caller does not choose one (ResNetRunner, engine.CaptionTrainer, the
# CNNEncoder_Atten shim, ConfigTrain.conv_mode, bench.py): the benchmarked mode.  "bf16x3" / "fp32" are the exact-operand alternatives.
DEFAULT_CONV_MODE = "f16x2"
L_COMPACT = 49          # distinct annotation cells when the 14x14 grid is a 2x2 replication of a 7x7 map (Q3)

# state_dict key  ->  field of dic_decoder_weights / dic_decoder_grads (include/dic.h)
DECODER_FIELDS = (
    ("attention.encoder_att.weight", "enc_att_w"), ("attention.encoder_att.bias", "enc_att_b"),
    ("attention.decoder_att.weight", "dec_att_w"), ("attention.decoder_att.bias", "dec_att_b"),
    ("attention.full_att.weight", "full_att_w"), ("attention.full_att.bias", "full_att_b"),
    ("embed.weight", "embed"),
    ("decode_step.weight_ih", "w_ih"), ("decode_step.weight_hh", "w_hh"),
    ("decode_step.bias_ih", "b_ih"), ("decode_step.bias_hh", "b_hh"),
    ("init_linear.weight", "init_w"), ("init_linear.bias", "init_b"),
    ("f_beta.weight", "fbeta_w"), ("f_beta.bias", "fbeta_b"),
    ("linear.weight", "out_w"), ("linear.bias", "out_b"),
)


class DecoderPtrs(C.Structure):
    """Mirrors dic_decoder_weights AND dic_decoder_grads (identical field order)."""
    _fields_ = [(name, C.c_void_p) for name in (
        "enc_att_w", "enc_att_b", "dec_att_w", "dec_att_b", "full_att_w", "full_att_b", "embed",
        "w_ih", "w_hh", "b_ih", "b_hh", "init_w", "init_b", "fbeta_w", "fbeta_b", "out_w", "out_b")]


def _dev_f32(t: torch.Tensor, what: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.DicError(f"{what}: tensor must live on the GPU (no CPU fallback)")
    if t.dtype != torch.float32:
        raise _lib.DicError(f"{what}: expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def decoder_ptrs(tensors: Dict[str, torch.Tensor]) -> Tuple[DecoderPtrs, list]:
    keep, s = [], DecoderPtrs()
    for key, field in DECODER_FIELDS:
        t = _dev_f32(tensors[key], key)
        keep.append(t)
        setattr(s, field, t.data_ptr())
    return s, keep


def _i32_host(values: Sequence[int]):
    arr = (C.c_int * len(values))(*[int(v) for v in values])
    return arr


@dataclass
class DecoderTape:
    """Everything dic_decoder_bwd needs from the matching forward call."""
    workspace: torch.Tensor
    dec_len: List[int]
    batch_sizes: List[int]
    n_packed: int
    tmax: int
    vocab: int
    captions: torch.Tensor
    drop_mult: Optional[torch.Tensor]
    mode: int
    temp: float
    alphas: torch.Tensor
    weights: Dict[str, torch.Tensor]
    cells: int = 196


def decoder_attention_relu_mask(tape: DecoderTape) -> torch.Tensor:
    """dic_decoder_inspect: which units of the attention ReLU passed in the forward that produced `tape`, bool
    [B, Tmax, cells, D_ATT] (rows of finished captions are meaningless).  For the parity tests' decision replay."""
    lib = _lib.load()
    B, T = len(tape.dec_len), tape.tmax
    dev = tape.workspace.device
    out = []
    for which, shape in ((1, (B, 1, tape.cells, D_ATT)), (2, (B, T, 1, D_ATT))):
        t = torch.empty(shape, dtype=torch.float32, device=dev)
        n = C.c_longlong(0)
        check(lib.dic_decoder_inspect(ptr(tape.workspace), C.c_size_t(tape.workspace.numel()), B, T, tape.vocab, tape.n_packed,
                                      tape.cells, which, ptr(t), C.byref(n), stream_ptr()), "dic_decoder_inspect")
        assert n.value == t.numel()
        out.append(t)
    return (out[0] + out[1]) > 0


def batch_sizes_of(dec_len: Sequence[int]) -> List[int]:
    return [sum(1 for l in dec_len if l > t) for t in range(max(dec_len))]


def decoder_forward(weights: Dict[str, torch.Tensor], feat_rgb: torch.Tensor, feat_depth: Optional[torch.Tensor],
                    captions: torch.Tensor, lengths: Sequence[int], drop_mult: Optional[torch.Tensor] = None,
                    mode: int = 0, gumbel_u: Optional[torch.Tensor] = None, temp: float = 1.0,
                    workspace: Optional[torch.Tensor] = None):
    """dic_decoder_fwd. Returns (logits_packed [N,V], alphas [B,Tmax,196], tape).
    Features of shape [B,49,2048] (the encoders' 7x7 maps before the 2x2 replication to 14x14) select the compact
    layout (dic_decoder_fwd_cells, soft attention only): same logits / alphas / gradients, 4x less feature traffic."""
    lib = _lib.load()
    B = feat_rgb.shape[0]
    cells = int(feat_rgb.shape[1])
    if cells not in (L_CELLS, L_COMPACT) or feat_rgb.shape[2] != D_ENC:
        raise _lib.DicError(f"features must be [B,{L_CELLS},{D_ENC}] or [B,{L_COMPACT},{D_ENC}], got {tuple(feat_rgb.shape)}")
    if cells == L_COMPACT and mode != 0:
        raise _lib.DicError("the compact 49-cell layout supports soft attention only")
    if feat_depth is not None and tuple(feat_depth.shape) != tuple(feat_rgb.shape):
        raise _lib.DicError("depth features must have the shape of the RGB features")
    dec_len = [int(l) - 1 for l in lengths]
    tmax = max(dec_len)
    bsz = batch_sizes_of(dec_len)
    n_packed = sum(bsz)
    vocab = weights["linear.weight"].shape[0]
    dev = feat_rgb.device
    wp, keep = decoder_ptrs(weights)
    f_rgb = _dev_f32(feat_rgb, "features")
    f_dep = _dev_f32(feat_depth, "depth_features") if feat_depth is not None else None
    caps = captions if captions.is_contiguous() else captions.contiguous()
    if caps.dtype != torch.int64 or not caps.is_cuda:
        raise _lib.DicError("captions must be an int64 GPU tensor")
    lib.dic_decoder_workspace_bytes.restype = C.c_size_t
    need = lib.dic_decoder_workspace_bytes(B, tmax, vocab, n_packed)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=dev)
    logits = torch.empty((n_packed, vocab), dtype=torch.float32, device=dev)
    alphas = torch.empty((B, tmax, L_CELLS), dtype=torch.float32, device=dev)
    dm = _dev_f32(drop_mult, "drop_mult") if drop_mult is not None else None
    gu = _dev_f32(gumbel_u, "gumbel_u") if gumbel_u is not None else None
    if cells == L_CELLS:
        rc = lib.dic_decoder_fwd(C.byref(wp), vocab, ptr(f_rgb), ptr(f_dep), ptr(caps), caps.stride(0),
                                 _i32_host(dec_len), B, ptr(dm), mode, ptr(gu), C.c_float(temp), ptr(logits), ptr(alphas),
                                 ptr(workspace), C.c_size_t(workspace.numel()), stream_ptr())
        check(rc, "dic_decoder_fwd")
    else:
        rc = lib.dic_decoder_fwd_cells(C.byref(wp), vocab, ptr(f_rgb), ptr(f_dep), cells, ptr(caps), caps.stride(0),
                                       _i32_host(dec_len), B, ptr(dm), ptr(logits), ptr(alphas), ptr(workspace),
                                       C.c_size_t(workspace.numel()), stream_ptr())
        check(rc, "dic_decoder_fwd_cells")
    tape = DecoderTape(workspace, dec_len, bsz, n_packed, tmax, vocab, caps, dm, mode, float(temp), alphas,
                       {k: t for (k, _), t in zip(DECODER_FIELDS, keep)}, cells)
    return logits, alphas, tape


def decoder_backward(tape: DecoderTape, dlogits: torch.Tensor, dalphas: Optional[torch.Tensor],
                     grads: Optional[Dict[str, torch.Tensor]] = None, want_dfeatures: bool = True):
    """dic_decoder_bwd. Returns (grads dict keyed like state_dict, d_features [B,196,2048] or None)."""
    lib = _lib.load()
    dev = dlogits.device
    B = len(tape.dec_len)
    if grads is None:
        grads = {k: torch.empty_like(t) for k, t in tape.weights.items()}
    gp, keep_g = decoder_ptrs(grads)
    wp, keep_w = decoder_ptrs(tape.weights)
    dfeat = torch.empty((B, tape.cells, D_ENC), dtype=torch.float32, device=dev) if want_dfeatures else None
    dl = _dev_f32(dlogits, "dlogits")
    da = _dev_f32(dalphas, "dalphas") if dalphas is not None else None
    if tape.cells == L_CELLS:
        rc = lib.dic_decoder_bwd(C.byref(wp), tape.vocab, ptr(tape.captions), tape.captions.stride(0),
                                 _i32_host(tape.dec_len), B, ptr(tape.drop_mult), tape.mode, C.c_float(tape.temp), ptr(dl),
                                 ptr(da), ptr(tape.alphas), C.byref(gp), ptr(dfeat), ptr(tape.workspace),
                                 C.c_size_t(tape.workspace.numel()), stream_ptr())
        check(rc, "dic_decoder_bwd")
    else:       # d_features is then the gradient w.r.t. the 7x7 maps
        rc = lib.dic_decoder_bwd_cells(C.byref(wp), tape.vocab, tape.cells, ptr(tape.captions), tape.captions.stride(0),
                                       _i32_host(tape.dec_len), B, ptr(tape.drop_mult), ptr(dl), ptr(da), ptr(tape.alphas),
                                       C.byref(gp), ptr(dfeat), ptr(tape.workspace), C.c_size_t(tape.workspace.numel()),
                                       stream_ptr())
        check(rc, "dic_decoder_bwd_cells")
    return grads, dfeat


def pack_targets(captions: torch.Tensor, lengths: Sequence[int]) -> torch.Tensor:
    lib = _lib.load()
    dec_len = [int(l) - 1 for l in lengths]
    n = sum(batch_sizes_of(dec_len))
    buf = torch.empty(n + (len(dec_len) + 1) // 2 + 2, dtype=torch.int64, device=captions.device)   # + int32 lengths
    caps = captions if captions.is_contiguous() else captions.contiguous()
    check(lib.dic_pack_targets(ptr(caps), caps.stride(0), _i32_host(dec_len), len(dec_len), ptr(buf), stream_ptr()),
          "dic_pack_targets")
    return buf[:n]


def caption_loss(logits: torch.Tensor, targets: torch.Tensor, alphas: Optional[torch.Tensor], lam: float = 0.7,
                 grad_scale: float = 1.0, in_place: bool = False, reg_grad_scale: Optional[float] = None):
    """dic_caption_loss. Returns (loss [1] device tensor, dlogits, dalphas or None).
    grad_scale multiplies dlogits, reg_grad_scale (default: the same value) multiplies dalphas - data parallel passes
    n_packed_r / sum n_packed and 1 / world (see include/dic.h)."""
    lib = _lib.load()
    n, v = logits.shape
    dev = logits.device
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    dlogits = logits if in_place else torch.empty_like(logits)
    B = alphas.shape[0] if alphas is not None else 0
    T = alphas.shape[1] if alphas is not None else 0
    dalphas = torch.empty_like(alphas) if alphas is not None else None
    scratch = torch.empty(n + B + 8, dtype=torch.float32, device=dev)
    rc = lib.dic_caption_loss(ptr(logits), ptr(targets), n, v, ptr(alphas), B, T, C.c_float(lam), C.c_float(grad_scale),
                              C.c_float(grad_scale if reg_grad_scale is None else reg_grad_scale),
                              ptr(loss), ptr(dlogits), ptr(dalphas), ptr(scratch), stream_ptr())
    check(rc, "dic_caption_loss")
    return loss, dlogits, dalphas


def _guard_ptr(word: Optional[torch.Tensor]):
    if word is None:
        return C.c_void_p(0)
    if not (word.is_cuda and word.dtype == torch.int32 and word.numel() >= 1 and word.is_contiguous()):
        raise _lib.DicError("the overflow guard word must be a contiguous int32 GPU tensor")
    return ptr(word)


def adamw_step(params: torch.Tensor, grads: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, step: int,
               lr: float = 1e-3, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
               weight_decay: float = 0.01, skip_if_raised: Optional[torch.Tensor] = None) -> None:
    """dic_adamw_step_guarded: `skip_if_raised` (int32 device word, optional) = the f16x2 overflow guard of the forward behind these
    gradients; when it is non-zero the kernel leaves parameters and moments untouched."""
    lib = _lib.load()
    for t in (params, grads, exp_avg, exp_avg_sq):
        if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
            raise _lib.DicError("adamw_step needs contiguous fp32 GPU buffers")
    rc = lib.dic_adamw_step_guarded(ptr(params), ptr(grads), ptr(exp_avg), ptr(exp_avg_sq), C.c_longlong(params.numel()), step,
                                    C.c_float(lr), C.c_float(beta1), C.c_float(beta2), C.c_float(eps), C.c_float(weight_decay),
                                    _guard_ptr(skip_if_raised), stream_ptr())
    check(rc, "dic_adamw_step_guarded")


def bn_ema_update(running: torch.Tensor, delta: torch.Tensor, momentum: float = 0.1,
                  skip_if_raised: Optional[torch.Tensor] = None) -> None:
    """dic_bn_ema_update_guarded: running = (1 - momentum) * running + delta (flat fp32 buffers of equal length), dropped on the
    device when the guard word `skip_if_raised` is non-zero."""
    if not (running.is_cuda and running.is_contiguous() and delta.is_contiguous() and running.numel() == delta.numel()):
        raise _lib.DicError("bn_ema_update needs two contiguous GPU buffers of equal length")
    check(_lib.load().dic_bn_ema_update_guarded(ptr(running), ptr(delta), C.c_longlong(running.numel()), C.c_float(momentum),
                                                _guard_ptr(skip_if_raised), stream_ptr()), "dic_bn_ema_update_guarded")


def dropout_mask(shape, p: float, seed: int, offset: int, device) -> torch.Tensor:
    lib = _lib.load()
    out = torch.empty(shape, dtype=torch.float32, device=device)
    rc = lib.dic_dropout_mask(ptr(out), C.c_longlong(out.numel()), C.c_float(p), C.c_uint64(seed), C.c_uint64(offset),
                              stream_ptr())
    check(rc, "dic_dropout_mask")
    return out


# ---------------------------------------------------------------------------------------------
# depth encoder (dic_depth_encoder_fwd / _bwd)
# ---------------------------------------------------------------------------------------------
DEPTH_FIELDS = tuple((f"{layer}.{kind}", f"{layer}_{'w' if kind == 'weight' else 'b'}")
                     for i in (1, 2, 3) for layer, kind in
                     ((f"conv{i}", "weight"), (f"conv{i}", "bias"), (f"bn{i}", "weight"), (f"bn{i}", "bias")))


class DepthPtrs(C.Structure):
    """Mirrors dic_depth_encoder_weights / dic_depth_encoder_grads."""
    _fields_ = [(f, C.c_void_p) for _, f in DEPTH_FIELDS]


class DepthBnState(C.Structure):
    _fields_ = [(f, C.c_void_p) for f in ("rm1", "rv1", "rm2", "rv2", "rm3", "rv3")]


def depth_ptrs(tensors: Dict[str, torch.Tensor]):
    keep, s = [], DepthPtrs()
    for key, field in DEPTH_FIELDS:
        t = _dev_f32(tensors[key], key)
        keep.append(t)
        setattr(s, field, t.data_ptr())
    return s, keep


@dataclass
class DepthTape:
    workspace: torch.Tensor
    depth: torch.Tensor
    weights: Dict[str, torch.Tensor]
    compact: bool = False


def depth_encoder_forward(weights: Dict[str, torch.Tensor], state: Dict[str, torch.Tensor], depth: torch.Tensor,
                          train: bool, workspace: Optional[torch.Tensor] = None, compact: bool = False):
    """dic_depth_encoder_fwd: depth [B,1,H,W] -> (features [B,196,2048], tape). `state` holds
    bn{1,2,3}.running_{mean,var} (updated in place when train)."""
    lib = _lib.load()
    d = _dev_f32(depth, "depth_map")
    B, c, H, W = d.shape
    if c != 1:
        raise _lib.DicError("depth map must be [B,1,H,W]")
    wp, keep = depth_ptrs(weights)
    st = DepthBnState()
    for i in (1, 2, 3):
        for short, name in (("rm", "running_mean"), ("rv", "running_var")):
            t = state[f"bn{i}.{name}"]
            if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
                raise _lib.DicError("BatchNorm running statistics must be contiguous fp32 GPU tensors")
            setattr(st, f"{short}{i}", t.data_ptr())
    lib.dic_depth_encoder_workspace_bytes.restype = C.c_size_t
    need = lib.dic_depth_encoder_workspace_bytes(B, H, W)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=d.device)
    if compact:          # the 7x7 map before the 2x2 replication (dic_depth_encoder_fwd_map; 224x224 inputs)
        if (H, W) != (224, 224):
            raise _lib.DicError("compact depth features need 224x224 inputs (a 7x7 final map)")
        out = torch.empty((B, L_COMPACT, D_ENC), dtype=torch.float32, device=d.device)
        rc = lib.dic_depth_encoder_fwd_map(C.byref(wp), C.byref(st), ptr(d), B, H, W, 1 if train else 0, ptr(out),
                                           ptr(workspace), C.c_size_t(workspace.numel()), stream_ptr())
        check(rc, "dic_depth_encoder_fwd_map")
    else:
        out = torch.empty((B, L_CELLS, D_ENC), dtype=torch.float32, device=d.device)
        rc = lib.dic_depth_encoder_fwd(C.byref(wp), C.byref(st), ptr(d), B, H, W, 1 if train else 0, ptr(out),
                                       ptr(workspace), C.c_size_t(workspace.numel()), stream_ptr())
        check(rc, "dic_depth_encoder_fwd")
    return out, DepthTape(workspace, d, {k: t for (k, _), t in zip(DEPTH_FIELDS, keep)}, compact)


def depth_status_word(tape: DepthTape) -> torch.Tensor:
    """int32[1] device view of the depth encoder's f16x2 overflow guard word (first 4 bytes of its workspace, include/dic.h): non-zero
    after a forward whose pooled activations left the fp16 range of the operand planes; the features were then filled with NaN."""
    return tape.workspace[:4].view(torch.int32)


def depth_encoder_backward(tape: DepthTape, d_features: torch.Tensor, grads: Optional[Dict[str, torch.Tensor]] = None):
    lib = _lib.load()
    if grads is None:
        grads = {k: torch.empty_like(t) for k, t in tape.weights.items()}
    gp, keep_g = depth_ptrs(grads)
    wp, keep_w = depth_ptrs(tape.weights)
    df = _dev_f32(d_features, "d_features")
    B, _, H, W = tape.depth.shape
    fn = lib.dic_depth_encoder_bwd_map if tape.compact else lib.dic_depth_encoder_bwd
    if tuple(df.shape) != (B, L_COMPACT if tape.compact else L_CELLS, D_ENC):
        raise _lib.DicError(f"d_features has shape {tuple(df.shape)}")
    rc = fn(C.byref(wp), ptr(tape.depth), ptr(df), B, H, W, C.byref(gp), ptr(tape.workspace),
            C.c_size_t(tape.workspace.numel()), stream_ptr())
    check(rc, "dic_depth_encoder_bwd")
    return grads


def depth_encoder_decisions(tape: DepthTape) -> Dict[str, torch.Tensor]:
    """dic_depth_encoder_inspect: the max-pool arg-max indices and pooled maps of the forward that produced `tape`
    (NHWC), for the parity tests' decision replay.  Keys: pooled1, argmax1, pooled2, argmax2, relu3."""
    lib = _lib.load()
    B, _, H, W = tape.depth.shape
    out = {}
    for which, name in ((1, "pooled1"), (2, "argmax1"), (3, "pooled2"), (4, "argmax2"), (5, "relu3")):
        n = C.c_longlong(0)
        check(lib.dic_depth_encoder_inspect(ptr(tape.workspace), C.c_size_t(tape.workspace.numel()), B, H, W, which,
                                            C.c_void_p(0), C.byref(n), stream_ptr()), "dic_depth_encoder_inspect")
        t = torch.empty(n.value, dtype=torch.float32 if which in (1, 3) else torch.uint8, device=tape.depth.device)
        check(lib.dic_depth_encoder_inspect(ptr(tape.workspace), C.c_size_t(tape.workspace.numel()), B, H, W, which,
                                            ptr(t), C.byref(n), stream_ptr()), "dic_depth_encoder_inspect")
        ch = 128 if which <= 2 else 512 if which <= 4 else 2048
        out[name] = t.view(B, -1, ch)          # [B, PH*PW, C]
    return out


# ---------------------------------------------------------------------------------------------
# RGB encoder (dic_resnet_fwd)
# ---------------------------------------------------------------------------------------------
class ConvBnLayer(C.Structure):
    _fields_ = [(f, C.c_void_p) for f in ("w", "gamma", "beta", "running_mean", "running_var", "w_hi", "w_mid", "w_lo")] + [("w_scale", C.c_float)]


CONV_MODES = {"fp32": 0, "bf16x3": 1, "f16x2": 2}


class ResNetRunner:
    """Holds the OHWI copies of the (frozen) ResNet conv weights and the layer table for dic_resnet_fwd.
    `tensors` is keyed like CNNEncoder_Atten.state_dict() ('backbone.0.weight', 'backbone.1.running_mean', ...)."""

    def __init__(self, tensors: Dict[str, torch.Tensor], layers: Sequence[int] = (3, 8, 36, 3), conv_mode: str = DEFAULT_CONV_MODE):
        """conv_mode "fp32": exact-fp32 MFMA convolutions; "bf16x3": fp32-accurate split-bf16 convolutions
        (each fp32 weight/activation = hi+mid+lo bf16 exactly, 6 products; csrc/gemm_bf3.hip); "f16x2": the same kernels on two
        fp16 planes of scaled values (3 products: half the matrix-core work, a few fp32 round-offs per product - inside the error
        envelope of an fp32 evaluation of the network; weights scaled per layer so that their largest magnitude lands in
        (2^13, 2^14] - scale = 2^floor(14 - log2 max|w|) -, activations by 4)."""
        from .synthetic import resnet152_spec
        lib = _lib.load()
        _lib.check_struct(lib, 0, ConvBnLayer)
        if conv_mode not in CONV_MODES:
            raise _lib.DicError(f"conv_mode must be one of {list(CONV_MODES)}")
        self.mode = CONV_MODES[conv_mode]
        self.blocks = (C.c_int * 4)(*[int(x) for x in layers])
        self.spec = resnet152_spec(layers)
        self.n_layers = len(self.spec)
        self.table = (ConvBnLayer * self.n_layers)()
        self.keep = []
        for i, (key, bn, co, ci, k, _s, _p) in enumerate(self.spec):
            w = _dev_f32(tensors[key], key)
            if tuple(w.shape) != (co, ci, k, k):
                raise _lib.DicError(f"{key}: expected {(co, ci, k, k)}, got {tuple(w.shape)}")
            if k == 1 or ci == 1:
                w_ohwi = w                               # same memory order
            else:
                w_ohwi = torch.empty_like(w)
                check(lib.dic_oihw_to_ohwi(ptr(w), ptr(w_ohwi), co, ci, k, k, stream_ptr()), "dic_oihw_to_ohwi")
            ent = self.table[i]
            ent.w = w_ohwi.data_ptr()
            tens = [w, w_ohwi]
            if self.mode == 2 and i == 0 and (co, ci, k) == (64, 3, 7):
                # 7x7 stem in the f16x2 format (round 4): two strip-ordered fp16 planes of scale * w (dic_resnet_pack_stem_weights_f16x2)
                import math
                wmax = float(w.abs().max())
                if not (wmax > 0.0 and math.isfinite(wmax)):
                    raise _lib.DicError(f"{key}: f16x2 mode needs finite, non-zero weights")
                scale = 2.0 ** math.floor(14 - math.log2(wmax))
                scratch = torch.empty(64 * 224, dtype=torch.float32, device=w.device)
                planes = [torch.empty(64 * 224, dtype=torch.int16, device=w.device) for _ in range(2)]
                check(lib.dic_resnet_pack_stem_weights_f16x2(ptr(w), ptr(scratch), ptr(planes[0]), ptr(planes[1]), C.c_float(scale),
                                                             stream_ptr()), "dic_resnet_pack_stem_weights_f16x2")
                ent.w_hi, ent.w_mid, ent.w_lo, ent.w_scale = planes[0].data_ptr(), planes[1].data_ptr(), None, scale
                tens += planes + [scratch]
            elif self.mode >= 1 and i == 0 and (co, ci, k) == (64, 3, 7):
                # 7x7 stem: strip-ordered weight planes [64][7][8][4] for the bf16x3 kernel (dic_resnet_pack_stem_weights)
                scratch = torch.empty(64 * 224, dtype=torch.float32, device=w.device)
                planes = [torch.empty(64 * 224, dtype=torch.int16, device=w.device) for _ in range(3)]
                check(lib.dic_resnet_pack_stem_weights(ptr(w), ptr(scratch), ptr(planes[0]), ptr(planes[1]), ptr(planes[2]),
                                                       stream_ptr()), "dic_resnet_pack_stem_weights")
                ent.w_hi, ent.w_mid, ent.w_lo = (pl.data_ptr() for pl in planes)
                tens += planes + [scratch]
            if self.mode == 1 and i > 0:             # (other C_in % 32 != 0 layers would stay on the exact-fp32 kernel)
                planes = [torch.empty(w_ohwi.numel(), dtype=torch.int16, device=w_ohwi.device) for _ in range(3)]
                check(lib.dic_split_bf16x3_paired(ptr(w_ohwi), C.c_longlong(co), ci * k * k, ptr(planes[0]),
                                                  ptr(planes[1]), ptr(planes[2]), stream_ptr()),
                      "dic_split_bf16x3_paired")
                ent.w_hi, ent.w_mid, ent.w_lo = (pl.data_ptr() for pl in planes)
                tens += planes
            if self.mode == 2 and i > 0:
                import math
                wmax = float(w_ohwi.abs().max())
                if not (wmax > 0.0 and math.isfinite(wmax)):
                    raise _lib.DicError(f"{key}: f16x2 mode needs finite, non-zero weights")
                scale = 2.0 ** math.floor(14 - math.log2(wmax))
                planes = [torch.empty(w_ohwi.numel(), dtype=torch.int16, device=w_ohwi.device) for _ in range(2)]
                check(lib.dic_split_f16x2_paired(ptr(w_ohwi), C.c_longlong(co), ci * k * k, C.c_float(scale), ptr(planes[0]),
                                                 ptr(planes[1]), stream_ptr()), "dic_split_f16x2_paired")
                ent.w_hi, ent.w_mid, ent.w_lo, ent.w_scale = planes[0].data_ptr(), planes[1].data_ptr(), None, scale
                tens += planes
            for field, name in (("gamma", "weight"), ("beta", "bias"), ("running_mean", "running_mean"),
                                ("running_var", "running_var")):
                t = tensors[bn + name]
                if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
                    raise _lib.DicError(f"{bn + name} must be a contiguous fp32 GPU tensor")
                setattr(ent, field, t.data_ptr())
                tens.append(t)
            self.keep.append(tens)
        self.workspace: Optional[torch.Tensor] = None
        self.train_forwards = 0          # train-mode forwards so far (BatchNorm num_batches_tracked, quirk Q1)

    def shadow(self, stats: Dict[str, torch.Tensor]) -> "ResNetRunner":
        """A second runner over the SAME frozen weights (tensors shared) with its own workspace and its own layer table whose
        BatchNorm running-statistic pointers are `stats[<bn prefix>running_mean / running_var]` - used by the engine to run
        several forwards ahead concurrently: each writes its running-statistic updates into its own zeroed scratch buffers
        (dic_bn_ema_update applies them later, in batch order)."""
        other = object.__new__(ResNetRunner)
        other.mode, other.blocks, other.spec, other.n_layers = self.mode, self.blocks, self.spec, self.n_layers
        other.table = (ConvBnLayer * self.n_layers)()
        for i, (_key, bn, *_rest) in enumerate(self.spec):
            other.table[i] = self.table[i]
            for field, name in (("running_mean", "running_mean"), ("running_var", "running_var")):
                t = stats[bn + name]
                if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
                    raise _lib.DicError(f"{bn + name} must be a contiguous fp32 GPU tensor")
                setattr(other.table[i], field, t.data_ptr())
        other.keep = [self.keep, list(stats.values())]
        other.workspace = None
        other.train_forwards = 0
        return other

    def forward(self, imgs: torch.Tensor, train_bn: bool, out: Optional[torch.Tensor] = None,
                compact: bool = False) -> torch.Tensor:
        """compact=True (224x224 inputs): returns the final 7x7 map [B,49,2048] itself instead of its 2x2 replication
        to [B,196,2048] (dic_resnet_fwd_map) - the input of the compact decoder layout."""
        lib = _lib.load()
        x = _dev_f32(imgs, "imgs")
        B, c, H, W = x.shape
        if c != 3:
            raise _lib.DicError("images must be [B,3,H,W]")
        lib.dic_resnet_workspace_bytes.restype = C.c_size_t
        need = lib.dic_resnet_workspace_bytes(B, H, W, self.blocks, self.mode)
        if self.workspace is None or self.workspace.numel() < need:
            self.workspace = torch.empty(need, dtype=torch.uint8, device=x.device)
        if compact and (H, W) != (224, 224):
            raise _lib.DicError("compact RGB features need 224x224 inputs (a 7x7 final map)")
        cells = L_COMPACT if compact else L_CELLS
        if out is None:
            out = torch.empty((B, cells, D_ENC), dtype=torch.float32, device=x.device)
        elif tuple(out.shape) != (B, cells, D_ENC):
            raise _lib.DicError(f"out must be {(B, cells, D_ENC)}, got {tuple(out.shape)}")
        fn = lib.dic_resnet_fwd_map if compact else lib.dic_resnet_fwd
        rc = fn(self.table, self.n_layers, self.blocks, ptr(x), B, H, W, 1 if train_bn else 0, self.mode, ptr(out),
                ptr(self.workspace), C.c_size_t(self.workspace.numel()), stream_ptr())
        check(rc, "dic_resnet_fwd")
        if train_bn and not torch.cuda.is_current_stream_capturing():
            self.train_forwards += 1
        return out

    # ---- f16x2 overflow guard (include/dic.h, dic_resnet_fwd): the status word is the first 4 bytes of the workspace -------------
    def status_word(self) -> torch.Tensor:
        """int32[1] device view of the status word of this runner's last forward (non-zero: an activation left the fp16 range of the
        f16x2 operand planes, or a non-finite value reached a convolution output; the features were filled with NaN)."""
        if self.workspace is None:
            raise _lib.DicError("ResNetRunner.status_word: no forward has run yet")
        return self.workspace[:4].view(torch.int32)

    def check_overflow(self) -> None:
        """Synchronising check of the last forward (one 4-byte read): raises DicError when its guard word is raised."""
        if self.mode == 2 and self.workspace is not None and int(self.status_word().item()) != 0:
            raise _lib.DicError("ResNet-152 forward in f16x2 arithmetic: an activation exceeded the fp16 range of the operand planes "
                                "(|x| > 16376) or a non-finite value reached a convolution - the features were filled with NaN and the "
                                "BatchNorm running statistics of the affected layers left untouched; use conv_mode='bf16x3' (exact "
                                "operands, no range limit) for these weights / inputs")


# ---------------------------------------------------------------------------------------------
# greedy decode + stand-alone attention
# ---------------------------------------------------------------------------------------------
def decoder_greedy(weights: Dict[str, torch.Tensor], feat_rgb: torch.Tensor, feat_depth: Optional[torch.Tensor],
                   id_start: int, max_length: int = 30, mode: int = 0, gumbel_u: Optional[torch.Tensor] = None):
    """dic_decoder_greedy. Returns (ids int64 [B,max_length] on device, alphas [B,max_length,196])."""
    lib = _lib.load()
    f_rgb = _dev_f32(feat_rgb, "features")
    f_dep = _dev_f32(feat_depth, "depth_features") if feat_depth is not None else None
    B = f_rgb.shape[0]
    vocab = weights["linear.weight"].shape[0]
    wp, keep = decoder_ptrs(weights)
    lib.dic_decoder_greedy_workspace_bytes.restype = C.c_size_t
    need = lib.dic_decoder_greedy_workspace_bytes(B, max_length, vocab)
    ws = torch.empty(need, dtype=torch.uint8, device=f_rgb.device)
    ids = torch.empty((B, max_length), dtype=torch.int64, device=f_rgb.device)
    alphas = torch.empty((B, max_length, L_CELLS), dtype=torch.float32, device=f_rgb.device)
    gu = _dev_f32(gumbel_u, "gumbel_u") if gumbel_u is not None else None
    rc = lib.dic_decoder_greedy(C.byref(wp), vocab, ptr(f_rgb), ptr(f_dep), B, C.c_longlong(int(id_start)), max_length,
                                mode, ptr(gu), ptr(ids), ptr(alphas), ptr(ws), C.c_size_t(ws.numel()), stream_ptr())
    check(rc, "dic_decoder_greedy")
    return ids, alphas


def attention_forward(att: Dict[str, torch.Tensor], feats: torch.Tensor, h: torch.Tensor, mode: int = 0,
                      gumbel_u: Optional[torch.Tensor] = None, temp: float = 1.0):
    """dic_attention_fwd. `att` holds encoder_att/decoder_att/full_att weight+bias. Returns (ctx [B,2048], alpha [B,196])."""
    lib = _lib.load()
    f = _dev_f32(feats, "encoder_out")
    hh = _dev_f32(h, "decoder_hidden")
    B = f.shape[0]
    if tuple(f.shape[1:]) != (L_CELLS, D_ENC) or tuple(hh.shape) != (B, D_HID):
        raise _lib.DicError("attention_forward: expected encoder_out [B,196,2048] and decoder_hidden [B,128]")
    t = {k: _dev_f32(v, k) for k, v in att.items()}
    lib.dic_attention_workspace_bytes.restype = C.c_size_t
    ws = torch.empty(lib.dic_attention_workspace_bytes(B), dtype=torch.uint8, device=f.device)
    ctx = torch.empty((B, D_ENC), dtype=torch.float32, device=f.device)
    alpha = torch.empty((B, L_CELLS), dtype=torch.float32, device=f.device)
    gu = _dev_f32(gumbel_u, "gumbel_u") if gumbel_u is not None else None
    rc = lib.dic_attention_fwd(ptr(t["encoder_att.weight"]), ptr(t["encoder_att.bias"]), ptr(t["decoder_att.weight"]),
                               ptr(t["decoder_att.bias"]), ptr(t["full_att.weight"]), ptr(t["full_att.bias"]), ptr(f),
                               ptr(hh), B, mode, ptr(gu), C.c_float(temp), ptr(ctx), ptr(alpha), ptr(ws),
                               C.c_size_t(ws.numel()), stream_ptr())
    check(rc, "dic_attention_fwd")
    return ctx, alpha


def attention_backward(att: Dict[str, torch.Tensor], feats: torch.Tensor, h: torch.Tensor, alpha: torch.Tensor,
                       d_ctx: torch.Tensor, d_alpha: Optional[torch.Tensor], mode: int = 0, temp: float = 1.0):
    """dic_attention_bwd. Returns (grads dict keyed like `att`, d_feats [B,196,2048], d_h [B,128])."""
    lib = _lib.load()
    f, hh, al, dc = _dev_f32(feats, "encoder_out"), _dev_f32(h, "decoder_hidden"), _dev_f32(alpha, "alpha"), _dev_f32(d_ctx, "d_ctx")
    da = _dev_f32(d_alpha, "d_alpha") if d_alpha is not None else None
    B = f.shape[0]
    t = {k: _dev_f32(v, k) for k, v in att.items()}
    g = {k: torch.empty_like(v) for k, v in t.items()}
    lib.dic_attention_bwd_workspace_bytes.restype = C.c_size_t
    ws = torch.empty(lib.dic_attention_bwd_workspace_bytes(B), dtype=torch.uint8, device=f.device)
    d_feats, d_h = torch.empty_like(f), torch.empty_like(hh)
    rc = lib.dic_attention_bwd(ptr(t["encoder_att.weight"]), ptr(t["encoder_att.bias"]), ptr(t["decoder_att.weight"]),
                               ptr(t["decoder_att.bias"]), ptr(t["full_att.weight"]), ptr(f), ptr(hh), ptr(al), B, mode,
                               C.c_float(temp), ptr(dc), ptr(da), ptr(g["encoder_att.weight"]), ptr(g["encoder_att.bias"]),
                               ptr(g["decoder_att.weight"]), ptr(g["decoder_att.bias"]), ptr(g["full_att.weight"]),
                               ptr(g["full_att.bias"]), ptr(d_feats), ptr(d_h), ptr(ws), C.c_size_t(ws.numel()), stream_ptr())
    check(rc, "dic_attention_bwd")
    return g, d_feats, d_h
