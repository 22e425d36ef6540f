"""Drop-in DPT_Depthestimator (Captioning_models/Depth_caption_model/DPT_model.py:16-67): the frozen DPT-Hybrid
monocular depth estimator that produces the depth maps in epoch 0 (depth_train.py:184-194).

Same surface as the reference class - DPT_Depthestimator(), .load_weight(), .standardize_depth_map(img), .forward(imgs) ->
[B,H,W], state_dict keys `model.<DPTDepthModel key>` - but no timm / torchvision: the forward is
depth_image_captioning_pub_amd.dpt.DptRunner (HIP kernels through the C ABI).  The reference's checkpoint
(omnidata_dpt_depth_v2.ckpt, an author-local absolute path, DPT_model.py:23) and timm's pretrained backbone are not
reachable offline, so construction draws procedural weights of the right shapes; a real checkpoint loads through
load_weight() / load_state_dict().  PARITY UNPINNED for the timm-defined pieces (see oracle/dpt_oracle.py)."""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import nn

from ... import synthetic as syn
from ..._lib import DicError
from ...dpt import DptRunner
from .. import util

image_size = 384                                        # DPT_model.py:14


class DPT_Depthestimator(nn.Module):
    def __init__(self, cfg: Optional[syn.DptConfig] = None, seed: int = 130):
        super().__init__()
        self.pretrained_weight_path = "/home/shirota/omnidata/torch/pretrained_models/omnidata_dpt_depth_v2.ckpt"   # :23
        self.cfg = cfg or syn.DptConfig()
        self._weights: Dict[str, torch.Tensor] = syn.dpt_weights(seed, self.cfg)
        self._runner: Optional[DptRunner] = None
        self.arith = "f16x2"     # DptRunner arithmetic: "f16x2" (default; raises if a layer input leaves the fp16 range), "bf16x3", "fp32"

    # ---- parameter plumbing (frozen: plain tensors, reference key names) --------------------------
    def _apply(self, fn, *args, **kwargs):               # .to(device) / .cuda() move the weight dict too
        self._weights = {k: fn(v) for k, v in self._weights.items()}
        self._runner = None
        return super()._apply(fn, *args, **kwargs)

    def state_dict(self, *args, **kwargs):
        return {"model." + k: v for k, v in self._weights.items()}

    def load_state_dict(self, state_dict, strict: bool = True):
        """Keys `model.<key>` as DPT_Depthestimator.state_dict() of the reference has them.  timm's classifier head
        (`model.pretrained.model.head.*`) is part of a real checkpoint but not of the depth path: ignored."""
        sd = {k[len("model."):]: v for k, v in state_dict.items() if k.startswith("model.")}
        sd = {k: v for k, v in sd.items() if not k.startswith("pretrained.model.head.")}
        missing = [k for k in self._weights if k not in sd]
        unexpected = [k for k in sd if k not in self._weights]
        if strict and (missing or unexpected):
            raise DicError(f"DPT_Depthestimator.load_state_dict: missing {missing[:5]}, unexpected {unexpected[:5]}")
        for k, v in sd.items():
            if k in self._weights:
                if tuple(v.shape) != tuple(self._weights[k].shape):
                    raise DicError(f"{k}: shape {tuple(v.shape)} vs {tuple(self._weights[k].shape)}")
                self._weights[k] = v.detach().to(self._weights[k].device, torch.float32).clone()
        self._runner = None

    def load_weight(self):
        """DPT_model.py:32-41: Lightning checkpoints keep the weights under 'state_dict' with a 6-character prefix."""
        checkpoint = torch.load(self.pretrained_weight_path, map_location="cpu", weights_only=True)
        if "state_dict" in checkpoint:
            checkpoint = {k[6:]: v for k, v in checkpoint["state_dict"].items()}
        self.load_state_dict({"model." + k: v for k, v in checkpoint.items()})

    # ---- the reference's methods -----------------------------------------------------------------------
    @torch.no_grad()
    def standardize_depth_map(self, img: torch.Tensor) -> torch.Tensor:
        """NaN -> 0.5, per-image min-max to [0,1]; img [B,1,H,W]   (DPT_model.py:43-61)."""
        return util.standardize_depth_map(img)

    @torch.no_grad()
    def forward(self, imgs: torch.Tensor) -> torch.Tensor:
        """[B,3,384,384] normalised with mean 0.5 / std 0.5 -> depth maps [B,384,384]   (DPT_model.py:63-67)."""
        if self._runner is None:
            self._runner = DptRunner(self._weights, self.cfg, arith=self.arith)
        return self._runner.forward(imgs)

    @torch.no_grad()
    def depth_maps_for_training(self, imgs_for_dep: torch.Tensor, out_size: int = 224) -> torch.Tensor:
        """The epoch-0 sequence of the training loop (depth_train.py:185-190): dpt(imgs) -> unsqueeze(1) ->
        standardize_depth_map -> T.Resize((224,224)) (bilinear; torchvision-version-dependent antialias default - the
        un-antialiased interpolation is used) -> [B,1,224,224], all on the device."""
        d = self.standardize_depth_map(self.forward(imgs_for_dep).unsqueeze(1))
        return util.resize_planes(d, out_size)
