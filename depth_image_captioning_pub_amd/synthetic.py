"""Procedural weights and inputs for the hot path.

No dataset, vocabulary pickle or pretrained checkpoint ships with the reference, so
every benchmark / parity input is generated on the host from documented NumPy PCG64
seeds (seed 123 echoes depth_main.py:7; rank r uses 123+r).  Weight *distributions*
follow the reference's initialisers: embed / linear.weight U(-0.1,0.1), linear.bias 0
(Depth_caption_model/depth_models.py:140-143), everything else PyTorch defaults;
ResNet-152 follows torchvision's init (Kaiming-normal fan-out convs, BN gamma=1 beta=0)
because IMAGENET1K_V2 weights are unreachable offline.

Everything returned is a CPU torch tensor keyed by the reference's state_dict names.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

L_CELLS, D_ENC, D_ATT, D_EMB, D_HID = 196, 2048, 128, 128, 128   # Captioning_models/config.py:11-15
IMAGENET_MEAN = (0.485, 0.456, 0.406)                             # Captioning_models/util.py:13
IMAGENET_STD = (0.229, 0.224, 0.225)
RESNET152_LAYERS = (3, 8, 36, 3)


def _rng(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(seed))


def _uniform(rng, shape, bound) -> torch.Tensor:
    return torch.from_numpy(rng.uniform(-bound, bound, size=shape).astype(np.float32))


def special_token_ids(vocab: int) -> Dict[str, int]:
    """<start>,<end>,<unk>,<null> are appended last, in that order (dataset/vocabulary_dict.ipynb cell 1)."""
    return {"<start>": vocab - 4, "<end>": vocab - 3, "<unk>": vocab - 2, "<null>": vocab - 1}


def decoder_weights(vocab: int, seed: int = 123, dim_attention: int = D_ATT, dim_embedding: int = D_EMB,
                    dim_encoder: int = D_ENC, dim_decoder: int = D_HID) -> Dict[str, torch.Tensor]:
    """17 tensors of CD_RNNDecoderWith{Soft,Hard}Attention (depth_models.py:106-143)."""
    r = _rng(seed)
    A, E, D, H, V = dim_attention, dim_embedding, dim_encoder, dim_decoder, vocab

    def lin(name, out_f, in_f):
        b = 1.0 / math.sqrt(in_f)
        return {name + ".weight": _uniform(r, (out_f, in_f), b), name + ".bias": _uniform(r, (out_f,), b)}

    w: Dict[str, torch.Tensor] = {}
    w.update(lin("attention.encoder_att", A, D))
    w.update(lin("attention.decoder_att", A, H))
    w.update(lin("attention.full_att", 1, A))
    w["embed.weight"] = _uniform(r, (V, E), 0.1)
    k = 1.0 / math.sqrt(H)
    w["decode_step.weight_ih"] = _uniform(r, (4 * H, E + D), k)
    w["decode_step.weight_hh"] = _uniform(r, (4 * H, H), k)
    w["decode_step.bias_ih"] = _uniform(r, (4 * H,), k)
    w["decode_step.bias_hh"] = _uniform(r, (4 * H,), k)
    w.update(lin("init_linear", 2 * H, D))
    w.update(lin("f_beta", D, H))
    w["linear.weight"] = _uniform(r, (V, H), 0.1)
    w["linear.bias"] = torch.zeros(V, dtype=torch.float32)
    return w


DEPTH_ENC_CONVS = (("conv1", 128, 1, 7), ("conv2", 512, 128, 3), ("conv3", 2048, 512, 1))


def depth_encoder_weights(seed: int = 124) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
    """(12 trainable tensors, BN running-stat state) of Depth_CNN_endoder (depth_models.py:19-24)."""
    r = _rng(seed)
    w: Dict[str, torch.Tensor] = {}
    st: Dict[str, torch.Tensor] = {}
    for i, (name, co, ci, k) in enumerate(DEPTH_ENC_CONVS, start=1):
        bound = 1.0 / math.sqrt(ci * k * k)
        w[f"{name}.weight"] = _uniform(r, (co, ci, k, k), bound)
        w[f"{name}.bias"] = _uniform(r, (co,), bound)
        w[f"bn{i}.weight"] = torch.ones(co)
        w[f"bn{i}.bias"] = torch.zeros(co)
        st[f"bn{i}.running_mean"] = torch.zeros(co)
        st[f"bn{i}.running_var"] = torch.ones(co)
    return w, st


def resnet152_spec(layers: Sequence[int] = RESNET152_LAYERS) -> List[Tuple[str, str, int, int, int, int, int]]:
    """[(conv_key, bn_prefix, c_out, c_in, k, stride, pad)] in execution order, keyed like
    CNNEncoder_Atten.backbone (Sequential of torchvision resnet children()[:-1])."""
    spec = [("backbone.0.weight", "backbone.1.", 64, 3, 7, 2, 3)]
    inplanes = 64
    for li, (nb, planes) in enumerate(zip(layers, (64, 128, 256, 512))):
        for bi in range(nb):
            p = f"backbone.{4 + li}.{bi}."
            stride = 2 if (li > 0 and bi == 0) else 1
            spec.append((p + "conv1.weight", p + "bn1.", planes, inplanes, 1, 1, 0))
            spec.append((p + "conv2.weight", p + "bn2.", planes, planes, 3, stride, 1))
            spec.append((p + "conv3.weight", p + "bn3.", planes * 4, planes, 1, 1, 0))
            if bi == 0:
                spec.append((p + "downsample.0.weight", p + "downsample.1.", planes * 4, inplanes, 1, stride, 0))
            inplanes = planes * 4
    return spec


def resnet152_weights(seed: int = 125, layers: Sequence[int] = RESNET152_LAYERS) -> Dict[str, torch.Tensor]:
    """Conv (OIHW) + BN tensors with torchvision's initialiser; running stats included (0 / 1)."""
    r = _rng(seed)
    w: Dict[str, torch.Tensor] = {}
    for key, bn, co, ci, k, _s, _p in resnet152_spec(layers):
        std = math.sqrt(2.0 / (co * k * k))                      # kaiming_normal_(mode="fan_out", relu)
        w[key] = torch.from_numpy((r.standard_normal((co, ci, k, k)) * std).astype(np.float32))
        w[bn + "weight"] = torch.ones(co)
        w[bn + "bias"] = torch.zeros(co)
        w[bn + "running_mean"] = torch.zeros(co)
        w[bn + "running_var"] = torch.ones(co)
    return w


def rgb_images(batch: int, seed: int = 123, size: int = 224) -> torch.Tensor:
    """U[0,1) then ImageNet normalisation (util.py:13,100)."""
    x = _rng(seed).random((batch, 3, size, size), dtype=np.float32)
    m = np.asarray(IMAGENET_MEAN, np.float32).reshape(1, 3, 1, 1)
    s = np.asarray(IMAGENET_STD, np.float32).reshape(1, 3, 1, 1)
    return torch.from_numpy((x - m) / s)


def raw_images(batch: int, seed: int = 123, size: int = 224) -> torch.Tensor:
    """The same U[0,1) draw as rgb_images, before any normalisation (what ToTensor() yields: util.py:100-101 then derive
    the ImageNet-normalised batch and the 384x384 copy for the depth estimator from it)."""
    return torch.from_numpy(_rng(seed).random((batch, 3, size, size), dtype=np.float32))


def depth_maps(batch: int, seed: int = 123, size: int = 224) -> torch.Tensor:
    """U[0,1) then per-image min-max to exactly [0,1] (mirrors DPT_model.py:50-59)."""
    x = _rng(seed + 1000).random((batch, 1, size, size), dtype=np.float32)
    lo = x.reshape(batch, -1).min(axis=1).reshape(batch, 1, 1, 1)
    hi = x.reshape(batch, -1).max(axis=1).reshape(batch, 1, 1, 1)
    return torch.from_numpy(((x - lo) / (hi - lo)).astype(np.float32))


def captions_fixed(batch: int, vocab: int, seq_len: int = 20, seed: int = 123) -> Tuple[torch.Tensor, List[int]]:
    """[B, seq_len+1]: <start>, seq_len-1 random word ids, <end>; all lengths equal (T = seq_len)."""
    ids = special_token_ids(vocab)
    c = _rng(seed + 2000).integers(0, vocab - 4, size=(batch, seq_len + 1), dtype=np.int64)
    c[:, 0] = ids["<start>"]
    c[:, -1] = ids["<end>"]
    return torch.from_numpy(c), [seq_len + 1] * batch


def captions_ragged(lengths: Sequence[int], vocab: int, seed: int = 123) -> Tuple[torch.Tensor, List[int]]:
    """Length-sorted (descending) ragged batch padded with <null> (util.py:95-108)."""
    lengths = sorted((int(l) for l in lengths), reverse=True)
    ids = special_token_ids(vocab)
    r = _rng(seed + 3000)
    c = np.full((len(lengths), lengths[0]), ids["<null>"], dtype=np.int64)
    for i, l in enumerate(lengths):
        c[i, :l] = r.integers(0, vocab - 4, size=l)
        c[i, 0] = ids["<start>"]
        c[i, l - 1] = ids["<end>"]
    return torch.from_numpy(c), lengths


def features(batch: int, seed: int, replicate: bool = True, scale: float = 1.0) -> torch.Tensor:
    """Non-negative annotation map [B,196,2048] shaped like an encoder output (post-ReLU);
    replicate=True makes each 2x2 block of the 14x14 grid identical (quirk Q3)."""
    r = _rng(seed)
    if replicate:
        g = np.maximum(r.standard_normal((batch, 7, 7, D_ENC)), 0).astype(np.float32) * scale
        g = np.repeat(np.repeat(g, 2, axis=1), 2, axis=2)
    else:
        g = np.maximum(r.standard_normal((batch, 14, 14, D_ENC)), 0).astype(np.float32) * scale
    return torch.from_numpy(np.ascontiguousarray(g.reshape(batch, L_CELLS, D_ENC)))


def dropout_multiplier(batch: int, tmax: int, p: float = 0.5, seed: int = 123, hidden: int = D_HID) -> torch.Tensor:
    """Explicit dropout multiplier (0 or 1/(1-p)) so CPU oracle and GPU path share the mask (Q6)."""
    keep = _rng(seed + 4000).random((batch, tmax, hidden)) >= p
    return torch.from_numpy((keep / (1.0 - p)).astype(np.float32))


def gumbel_uniforms(tmax: int, batch: int, seed: int = 123) -> torch.Tensor:
    """u ~ U(0,1) [T,B,196] for the hard path; explicit input (Q6, attention.py:17)."""
    u = _rng(seed + 5000).random((tmax, batch, L_CELLS), dtype=np.float32)
    return torch.from_numpy(np.clip(u, 1e-6, 1.0 - 1e-6).astype(np.float32))


# ------------------------------------------------------------------------------------------------
# DPT-Hybrid depth front-end (BASELINE config 5): DPTDepthModel(backbone='vitb_rn50_384')
#   (Depth_caption_model/modules/midas/dpt_depth.py:26-107, blocks.py:49-75,231-341, vit.py:345-477) over timm 0.4.12's
#   vit_base_resnet50_384.  The checkpoint (DPT_model.py:23) and the timm weights are unreachable offline: procedural
#   weights with the state_dict names and shapes DPTDepthModel.state_dict() has.
# ------------------------------------------------------------------------------------------------
class DptConfig:
    """Architecture sizes.  Default = vitb_rn50_384; tests shrink `layers` / `depth` (hooks follow) to keep the oracle fast."""

    def __init__(self, layers=(3, 4, 9), depth=12, hooks=(8, 11), heads=12, embed=768, mlp=3072, features=256,
                 pos_grid=24, stem=64, channels=(256, 512, 1024)):
        self.layers, self.depth, self.hooks, self.heads = tuple(layers), depth, tuple(hooks), heads
        self.embed, self.mlp, self.features, self.pos_grid = embed, mlp, features, pos_grid
        self.stem, self.channels = stem, tuple(channels)


def dpt_stage_spec(cfg: "DptConfig"):
    """[(prefix, in_chs, mid_chs, out_chs, stride, has_downsample)] of the ResNetV2 bottlenecks in execution order."""
    spec, prev = [], cfg.stem
    for s, (nb, out) in enumerate(zip(cfg.layers, cfg.channels)):
        for b in range(nb):
            p = f"pretrained.model.patch_embed.backbone.stages.{s}.blocks.{b}."
            spec.append((p, prev, out // 4, out, (1 if s == 0 else 2) if b == 0 else 1, b == 0))
            prev = out
    return spec


def dpt_weights(seed: int = 130, cfg: "DptConfig" = None) -> Dict[str, torch.Tensor]:
    cfg = cfg or DptConfig()
    r = _rng(seed)
    w: Dict[str, torch.Tensor] = {}

    def normal(shape, std):
        return torch.from_numpy((r.standard_normal(shape) * std).astype(np.float32))

    def conv(key, co, ci, k, bias=True):
        bound = 1.0 / math.sqrt(ci * k * k)                       # nn.Conv2d default initialiser
        w[key + ".weight"] = _uniform(r, (co, ci, k, k), bound * math.sqrt(3.0))
        if bias:
            w[key + ".bias"] = _uniform(r, (co,), bound)

    def norm(key, c):
        w[key + ".weight"] = torch.from_numpy((1.0 + 0.1 * r.standard_normal(c)).astype(np.float32))
        w[key + ".bias"] = torch.from_numpy((0.1 * r.standard_normal(c)).astype(np.float32))

    def linear(key, out_f, in_f, std=0.02):
        w[key + ".weight"] = normal((out_f, in_f), std)
        w[key + ".bias"] = normal((out_f,), 0.02)

    bb = "pretrained.model.patch_embed.backbone."
    w[bb + "stem.conv.weight"] = normal((cfg.stem, 3, 7, 7), 0.1)           # StdConv2dSame: standardised at use
    norm(bb + "stem.norm", cfg.stem)
    for p, cin, mid, out, _stride, ds in dpt_stage_spec(cfg):
        if ds:
            w[p + "downsample.conv.weight"] = normal((out, cin, 1, 1), 0.1)
            norm(p + "downsample.norm", out)
        w[p + "conv1.weight"] = normal((mid, cin, 1, 1), 0.1)
        norm(p + "norm1", mid)
        w[p + "conv2.weight"] = normal((mid, mid, 3, 3), 0.1)
        norm(p + "norm2", mid)
        w[p + "conv3.weight"] = normal((out, mid, 1, 1), 0.1)
        norm(p + "norm3", out)
    P = "pretrained.model."
    conv(P + "patch_embed.proj", cfg.embed, cfg.channels[-1], 1)
    w[P + "cls_token"] = normal((1, 1, cfg.embed), 0.02)
    w[P + "pos_embed"] = normal((1, 1 + cfg.pos_grid * cfg.pos_grid, cfg.embed), 0.02)
    for i in range(cfg.depth):
        b = P + f"blocks.{i}."
        norm(b + "norm1", cfg.embed)
        linear(b + "attn.qkv", 3 * cfg.embed, cfg.embed, std=0.05)
        linear(b + "attn.proj", cfg.embed, cfg.embed)
        norm(b + "norm2", cfg.embed)
        linear(b + "mlp.fc1", cfg.mlp, cfg.embed)
        linear(b + "mlp.fc2", cfg.embed, cfg.mlp)
    norm(P + "norm", cfg.embed)
    for n in (3, 4):
        linear(f"pretrained.act_postprocess{n}.0.project.0", cfg.embed, 2 * cfg.embed)
        conv(f"pretrained.act_postprocess{n}.3", cfg.embed, cfg.embed, 1)
    conv("pretrained.act_postprocess4.4", cfg.embed, cfg.embed, 3)
    F_ = cfg.features
    for n, cin in zip((1, 2, 3, 4), (cfg.channels[0], cfg.channels[1], cfg.embed, cfg.embed)):
        conv(f"scratch.layer{n}_rn", F_, cin, 3, bias=False)
        rn = f"scratch.refinenet{n}."
        conv(rn + "out_conv", F_, F_, 1)
        for u in (1, 2):
            conv(rn + f"resConfUnit{u}.conv1", F_, F_, 3)
            conv(rn + f"resConfUnit{u}.conv2", F_, F_, 3)
    conv("scratch.output_conv.0", F_ // 2, F_, 3)
    conv("scratch.output_conv.2", 32, F_ // 2, 3)
    conv("scratch.output_conv.4", 1, 32, 1)
    w["scratch.output_conv.4.bias"] = torch.full((1,), 0.3)       # keeps the final ReLU from clipping most of the map
    return w


def dpt_images(batch: int, seed: int = 123, size: int = 384) -> torch.Tensor:
    """What util.dep_trans hands the estimator: RGB in [0,1) -> Normalize(0.5, 0.5) = [-1,1)   (util.py:14-17)."""
    x = _rng(seed + 6000).random((batch, 3, size, size), dtype=np.float32)
    return torch.from_numpy((x - 0.5) / 0.5)
