"""Drop-in RGB encoder and base (no-depth) decoders
(Captioning_models/Base_caption_model/base_caption_models.py:13-45, 49-250, 257-508).

CNNEncoder_Atten keeps the reference's parameter tree - `backbone` = Sequential of the torchvision
ResNet-152 children()[:-1] with avgpool -> AdaptiveAvgPool2d(14) - so its state_dict keys
(`backbone.0.weight`, `backbone.4.0.conv1.weight`, `backbone.1.running_mean`, ...) match and a torchvision
IMAGENET1K_V2 checkpoint loads with load_state_dict.  torchvision itself is not needed: the forward runs
dic_resnet_fwd in the arithmetic `conv_mode` names (default native.DEFAULT_CONV_MODE = "f16x2", the benchmarked
mode: fp32 operands as two fp16 planes, fp32-level results, guarded against the format's range limit - see
forward(); "bf16x3" / "fp32" = exact operands).  No pretrained weights are reachable offline, so construction
uses torchvision's initialiser (Kaiming-normal fan-out, BN gamma=1 beta=0).
"""
from __future__ import annotations

import torch
from torch import nn

from ... import native
from ..._lib import DicError
from ..Depth_caption_model.depth_models import _CaptionDecoderBase

_LAYERS = (3, 8, 36, 3)


class _Bottleneck(nn.Module):
    """Parameter holder with torchvision's Bottleneck attribute names (conv1/bn1/conv2/bn2/conv3/bn3/downsample)."""
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int, downsample: bool):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)   # v1.5: stride on the 3x3
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))


def _make_stage(inplanes: int, planes: int, blocks: int, stride: int) -> nn.Sequential:
    mods = [_Bottleneck(inplanes, planes, stride, True)]
    mods += [_Bottleneck(planes * 4, planes, 1, False) for _ in range(1, blocks)]
    return nn.Sequential(*mods)


class CNNEncoder_Atten(nn.Module):
    def __init__(self, encoded_img_size: int, layers=_LAYERS, conv_mode: str = None):
        super().__init__()
        self.conv_mode = conv_mode or native.DEFAULT_CONV_MODE      # (not part of the reference's signature: keyword only in practice)
        self.check_overflow = True     # f16x2: read the guard word after every forward (one 4-byte device read = a host sync, as
                                       # the reference's own loop does with loss.item() every iteration, depth_train.py:224)
        if encoded_img_size != 14:
            raise DicError("the native path is built for the reference's 14x14 annotation grid")
        self._layers = tuple(layers)
        stem = [nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
                nn.MaxPool2d(3, stride=2, padding=1)]
        stages, inpl = [], 64
        for i, (planes, nb) in enumerate(zip((64, 128, 256, 512), self._layers)):
            stages.append(_make_stage(inpl, planes, nb, 1 if i == 0 else 2))
            inpl = planes * 4
        self.backbone = nn.Sequential(*stem, *stages, nn.AdaptiveAvgPool2d(encoded_img_size))
        for m in self.modules():                                   # torchvision's ResNet initialiser
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        for p in self.parameters():                                # frozen: never handed to the optimiser
            p.requires_grad_(False)                                # (depth_train.py:136) and run under no_grad
        self._runner = None
        self._runner_key = None

    def _native(self) -> native.ResNetRunner:
        sd = {"backbone." + k: v for k, v in self.backbone.state_dict(keep_vars=True).items()
              if not k.endswith("num_batches_tracked")}
        key = tuple((v.data_ptr(), v._version) for k, v in sd.items() if k.endswith("weight") and v.dim() == 4)
        key = key + (self.conv_mode,)
        if self._runner is None or key != self._runner_key:
            self._runner = native.ResNetRunner({k: v.detach() for k, v in sd.items()}, self._layers, conv_mode=self.conv_mode)
            self._runner_key = key
        return self._runner

    @torch.no_grad()
    def forward(self, imgs: torch.Tensor) -> torch.Tensor:
        """[B,3,H,W] -> [B,196,2048].  train() mode normalises with batch statistics and updates the running
        statistics although the weights are frozen (quirk Q1, depth_train.py:161); eval() uses running stats.
        In f16x2 arithmetic an activation beyond +-16376 does not fit the operand planes: the library then raises its guard
        word and fills the features with NaN, and this method raises DicError (check_overflow = False leaves the check to the
        caller: runner.check_overflow() or the NaN features)."""
        runner = self._native()
        out = runner.forward(imgs, train_bn=self.training)
        if self.check_overflow:
            runner.check_overflow()
        if self.training:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.num_batches_tracked += 1
        return out


class RNNDecoderWithSoftAttention(_CaptionDecoderBase):
    """base-soft decoder: identical to the CD_ variant without the depth features (base_caption_models.py:49-250)."""

    def __init__(self, dim_attention: int, dim_embedding: int, dim_encoder: int, dim_decoder: int, vocab_size: int,
                 dropout: float = 0.5):
        super().__init__()
        self._build(dim_attention, dim_embedding, dim_encoder, dim_decoder, vocab_size, dropout)

    def forward(self, features: torch.Tensor, captions: torch.Tensor, lengths: list):
        return self._run(features, None, captions, lengths, mode=0)

    def sample(self, features, word_to_id, max_length=30):
        return super().sample(features, None, word_to_id, max_length)

    def batch_sample(self, features, word_to_id, max_length=30):
        return super().batch_sample(features, None, word_to_id, max_length)


class RNNDecoderWithHardAttention(_CaptionDecoderBase):
    """base-hard decoder (base_caption_models.py:257-508)."""
    hard = True

    def __init__(self, dim_attention: int, dim_embedding: int, dim_encoder: int, dim_decoder: int, vocab_size: int,
                 device: str, dropout: float = 0.5):
        super().__init__()
        self.device = device
        self._build(dim_attention, dim_embedding, dim_encoder, dim_decoder, vocab_size, dropout)

    def forward(self, features, captions, lengths, temp):
        packed, _ = self._run(features, None, captions, lengths, mode=1, temp=float(temp))
        return packed

    @torch.no_grad()
    def eval_forward(self, features, captions, lengths):
        packed, _ = self._run(features, None, captions, lengths, mode=2)
        return packed

    def sample(self, features, word_to_id, max_length=30):
        return super().sample(features, None, word_to_id, max_length)

    def batch_sample(self, features, word_to_id, max_length=30):
        return super().batch_sample(features, None, word_to_id, max_length)
