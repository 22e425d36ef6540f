"""Host data path with the reference's function names (Captioning_models/util.py:52-221): tokenizer and the two
collate functions.  The collate functions are HOST-ONLY (strings, stacking, sorting, padding; CPU tensors out): the
reference hands them to a DataLoader with num_workers=4 (depth_train.py:93, config.py:65) and forked workers cannot
touch the GPU.  The tensor work the reference does inside its collate - ImageNet normalisation and the 384x384 bilinear
copy for the depth estimator (util.py:100-101) - runs in libdic_hip.so (csrc/data_ops.hip) once the batch has arrived in
the training process: `device_transforms(raw_imgs.to(device))`; likewise per-image depth standardisation and the
depth-cache lookup."""
from __future__ import annotations

import ctypes as C
import random
from typing import Dict, List, Sequence, Tuple, Union

import torch

from .. import _lib
from .._lib import check, ptr, stream_ptr

image_size = 384                                   # util.py:12
IMAGENET_MEAN = (0.485, 0.456, 0.406)              # util.py:13
IMAGENET_STD = (0.229, 0.224, 0.225)


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def norm_trans(imgs: torch.Tensor) -> torch.Tensor:
    """T.Normalize(ImageNet mean/std) on a GPU batch [B,3,H,W]   (util.py:13)."""
    if not imgs.is_cuda:
        raise _lib.DicError("norm_trans: the batch must be on the GPU (no CPU fallback)")
    x = imgs.contiguous()
    out = torch.empty_like(x)
    B, Cc, H, W = x.shape
    check(_lib.load().dic_normalize_images(ptr(x), ptr(out), B, Cc, H, W, _f3(IMAGENET_MEAN), _f3(IMAGENET_STD),
                                           stream_ptr()), "dic_normalize_images")
    return out


def dep_trans(imgs: torch.Tensor) -> torch.Tensor:
    """T.Resize(384, bilinear) + T.CenterCrop(384) + T.Normalize(0.5, 0.5) for the depth estimator (util.py:14-17)."""
    if not imgs.is_cuda:
        raise _lib.DicError("dep_trans: the batch must be on the GPU (no CPU fallback)")
    x = imgs.contiguous()
    B, Cc, H, W = x.shape
    out = torch.empty((B, Cc, image_size, image_size), dtype=torch.float32, device=x.device)
    check(_lib.load().dic_resize_bilinear(ptr(x), B * Cc, H, W, image_size, image_size, C.c_float(2.0), C.c_float(-1.0),
                                          ptr(out), stream_ptr()), "dic_resize_bilinear")
    return out


def resize_planes(x: torch.Tensor, size: int) -> torch.Tensor:
    """T.Resize((size, size)) of a square GPU tensor [B,C,S,S], bilinear, align_corners False, no antialias
    (depth_transforms of depth_train.py:67,190)."""
    if not x.is_cuda or x.shape[-1] != x.shape[-2]:
        raise _lib.DicError("resize_planes: square GPU tensors only")
    x = x.contiguous()
    B, Cc, H, W = x.shape
    out = torch.empty((B, Cc, size, size), dtype=torch.float32, device=x.device)
    check(_lib.load().dic_resize_bilinear(ptr(x), B * Cc, H, W, size, size, C.c_float(1.0), C.c_float(0.0), ptr(out),
                                          stream_ptr()), "dic_resize_bilinear")
    return out


def standardize_depth_map(depth: torch.Tensor) -> torch.Tensor:
    """Per-image min-max to [0,1], NaN -> 0.5 first (DPT_model.py:43-61). depth: [B,1,H,W] on the GPU."""
    d = depth.contiguous().clone()
    B = d.shape[0]
    check(_lib.load().dic_depth_standardize(ptr(d), B, C.c_longlong(d.numel() // B), stream_ptr()),
          "dic_depth_standardize")
    return d


def _clean_tokens(caption: str) -> List[str]:
    out = []
    for token in caption.lower().split():              # util.py:119-130
        if token in (".", ","):
            continue
        out.append(token.rstrip(".").rstrip(","))
    return out


def tokenize_caption(caption: str, word_to_id: Dict[str, int]) -> torch.Tensor:
    """'<start>' + words + '<end>' -> ids, unknown words -> '<unk>' (util.py:118-143). Like the reference this returns
    a float tensor (torch.Tensor(list)); the collate functions cast when they fill the int64 batch."""
    unk = word_to_id["<unk>"]
    ids = [word_to_id.get(k, unk) for k in ["<start>"] + _clean_tokens(caption) + ["<end>"]]
    return torch.Tensor(ids)


def untokenize_caption(caption: str, word_to_id: Dict[str, int]) -> str:
    """Lower-cased caption with out-of-vocabulary words replaced by '<unk>' (util.py:145-166)."""
    return " ".join(w if w in word_to_id else "<unk>" for w in _clean_tokens(caption))


def _pad_batch(captions: Sequence[torch.Tensor], null_id: int):
    lengths = [int(c.shape[0]) for c in captions]
    targets = torch.full((len(captions), max(lengths)), null_id, dtype=torch.int64)       # util.py:103-108
    for i, cap in enumerate(captions):
        targets[i, : lengths[i]] = cap[: lengths[i]]
    return targets, lengths


def collate_func(batch: Sequence[Tuple[Union[torch.Tensor, Sequence[str]]]], word_to_id: Dict[str, int]):
    """(imgs, targets, lengths): one of the 5 captions at random, sorted by length (descending), padded with
    '<null>' (util.py:57-78).  Host only, CPU tensors; like the reference this collate does not normalise the images
    (base_train.py does that in its T.Compose)."""
    imgs, captions = zip(*batch)
    captions = [tokenize_caption(random.choice(cap), word_to_id) for cap in captions]
    order = sorted(range(len(captions)), key=lambda i: len(captions[i]), reverse=True)
    imgs = torch.stack([imgs[i] for i in order])
    targets, lengths = _pad_batch([captions[i] for i in order], word_to_id["<null>"])
    return imgs, targets, lengths


def collate_func_for_dep(batch: Sequence[Tuple[Union[torch.Tensor, Sequence[str]]]], word_to_id: Dict[str, int]):
    """(imgs, imgs_for_dep, targets, lengths, allcaps) with the reference's arity and order (util.py:80-110), host only.
    Both image slots hold the SAME un-normalised CPU stack [B,3,H,W] in [0,1]: the reference's two transforms
    (util.py:100-101) are applied on the GPU by the training loop, `imgs, imgs_for_dep = device_transforms(raw.to(dev))`."""
    imgs, captions = zip(*batch)
    allcaps = [" ".join(cap) for cap in captions]
    captions = [tokenize_caption(random.choice(cap), word_to_id) for cap in captions]
    order = sorted(range(len(captions)), key=lambda i: len(captions[i]), reverse=True)
    raw = torch.stack([imgs[i] for i in order])
    targets, lengths = _pad_batch([captions[i] for i in order], word_to_id["<null>"])
    return raw, raw, targets, lengths, [allcaps[i] for i in order]


def device_transforms(raw_imgs: torch.Tensor):
    """norm_trans + dep_trans of a raw GPU batch (util.py:13-17,100-101) -> (imgs [B,3,H,W], imgs_for_dep [B,3,384,384])."""
    return norm_trans(raw_imgs), dep_trans(raw_imgs)


class DepthCache:
    """Device-resident replacement of the reference's `depth_dic` (depth_train.py:192-202): depth maps predicted in
    epoch 0 are kept in one HBM table keyed by the joined caption string; later epochs fetch a whole batch with one
    row-gather kernel instead of B host->device copies and torch.cat calls."""

    def __init__(self, capacity: int, height: int = 224, width: int = 224, device: str = "cuda:0"):
        self.table = torch.empty((capacity, 1, height, width), dtype=torch.float32, device=device)
        self.slot: Dict[str, int] = {}

    def put(self, keys: Sequence[str], depth_maps: torch.Tensor) -> None:
        for i, k in enumerate(keys):
            s = self.slot.get(k)
            if s is None:
                if len(self.slot) >= self.table.shape[0]:
                    raise _lib.DicError("DepthCache is full")          # (checked before the key is registered)
                s = self.slot[k] = len(self.slot)
            self.table[s].copy_(depth_maps[i])

    def get(self, keys: Sequence[str]) -> torch.Tensor:
        idx = torch.tensor([self.slot[k] for k in keys], dtype=torch.int64, device=self.table.device)
        out = torch.empty((len(keys),) + tuple(self.table.shape[1:]), dtype=torch.float32, device=self.table.device)
        row = self.table[0].numel()
        check(_lib.load().dic_gather_rows(ptr(self.table), ptr(idx), len(keys), C.c_longlong(row), ptr(out),
                                          stream_ptr()), "dic_gather_rows")
        return out
