"""Soft / hard (Gumbel) attention modules with the reference's call surface
(Captioning_models/attention.py:6-167).  Parameters are ordinary nn.Linear sub-modules (so
state_dict keys match: encoder_att / decoder_att / full_att); forward runs dic_attention_fwd.

Inside the decoders the attention is fused into the step kernels (the decoders read these sub-modules'
parameters).  The stand-alone forward is an ordinary differentiable module like the reference's: a
torch.autograd.Function wraps dic_attention_fwd / dic_attention_bwd (gradients for the six parameters, encoder_out and
decoder_hidden); Hard_sample (Gumbel-max) is not differentiable in the reference either."""
from __future__ import annotations

import torch
from torch import nn

from .. import native


def _att_tensors(mod: nn.Module):
    return {"encoder_att.weight": mod.encoder_att.weight, "encoder_att.bias": mod.encoder_att.bias,
            "decoder_att.weight": mod.decoder_att.weight, "decoder_att.bias": mod.decoder_att.bias,
            "full_att.weight": mod.full_att.weight, "full_att.bias": mod.full_att.bias}


def _detached(d):
    return {k: v.detach() for k, v in d.items()}


_ATT_KEYS = ("encoder_att.weight", "encoder_att.bias", "decoder_att.weight", "decoder_att.bias", "full_att.weight",
             "full_att.bias")


class _AttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, encoder_out, decoder_hidden, *params):
        att = {k: p.detach() for k, p in zip(_ATT_KEYS, params)}
        enc, hid = encoder_out.detach().contiguous(), decoder_hidden.detach().contiguous()
        c, alpha = native.attention_forward(att, enc, hid, mode=cfg["mode"], gumbel_u=cfg.get("gumbel_u"),
                                            temp=cfg.get("temp", 1.0))
        ctx.cfg, ctx.att = cfg, att
        ctx.save_for_backward(enc, hid, alpha)
        return c, alpha

    @staticmethod
    def backward(ctx, d_ctx, d_alpha):
        enc, hid, alpha = ctx.saved_tensors
        if d_ctx is None:
            d_ctx = torch.zeros((enc.shape[0], enc.shape[2]), dtype=torch.float32, device=enc.device)
        g, d_enc, d_h = native.attention_backward(ctx.att, enc, hid, alpha, d_ctx.contiguous(),
                                                  d_alpha.contiguous() if d_alpha is not None else None,
                                                  mode=ctx.cfg["mode"], temp=ctx.cfg.get("temp", 1.0))
        return (None, d_enc, d_h) + tuple(g[k] for k in _ATT_KEYS)


def _apply(mod: nn.Module, cfg, encoder_out, decoder_hidden):
    t = _att_tensors(mod)
    return _AttentionFn.apply(cfg, encoder_out, decoder_hidden, *[t[k] for k in _ATT_KEYS])


class Gumbel_softmax(nn.Module):
    """attention.py:6-48. The uniform draw comes from the CPU generator exactly like the reference
    (torch.rand(batch_size, k), attention.py:17,40) and is handed to the kernel as an input."""

    def __init__(self, k):
        super().__init__()
        self.k = k

    def draw(self, batch_size: int, device) -> torch.Tensor:
        return torch.rand(batch_size, self.k).to(device)


class Soft_Attention(nn.Module):
    def __init__(self, dim_encoder: int, dim_decoder: int, dim_attention: int):
        super().__init__()
        self.encoder_att = nn.Linear(dim_encoder, dim_attention)
        self.decoder_att = nn.Linear(dim_decoder, dim_attention)
        self.full_att = nn.Linear(dim_attention, 1)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, encoder_out: torch.Tensor, decoder_hidden: torch.Tensor):
        """-> (context_vector [B,2048], alpha [B,196])   (attention.py:81-95); differentiable"""
        return _apply(self, {"mode": 0}, encoder_out, decoder_hidden)


class Hard_Attention(nn.Module):
    def __init__(self, dim_encoder: int, dim_decoder: int, dim_attention: int, k=196):
        super().__init__()
        self.encoder_att = nn.Linear(dim_encoder, dim_attention)
        self.decoder_att = nn.Linear(dim_decoder, dim_attention)
        self.full_att = nn.Linear(dim_attention, 1)
        self.relu = nn.ReLU(inplace=True)
        self.gumbel_softmax = Gumbel_softmax(k)

    def forward(self, encoder_out, decoder_hidden, device, temp):
        """Gumbel-softmax attention (attention.py:132-148); differentiable (the noise is a constant of the graph)."""
        u = self.gumbel_softmax.draw(encoder_out.shape[0], encoder_out.device)
        return _apply(self, {"mode": 1, "gumbel_u": u, "temp": float(temp)}, encoder_out, decoder_hidden)

    @torch.no_grad()
    def Hard_sample(self, encoder_out, decoder_hidden, device):
        """Gumbel-max one-hot attention (attention.py:150-167); alpha comes back as int64 one-hot."""
        u = self.gumbel_softmax.draw(encoder_out.shape[0], encoder_out.device)
        ctx, alpha = native.attention_forward(_detached(_att_tensors(self)), encoder_out, decoder_hidden, mode=2,
                                              gumbel_u=u)
        return ctx, alpha.to(torch.int64)
