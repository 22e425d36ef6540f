"""Training harness with the reference's entry points train_Cdepth_soft(ext, useData) /
train_Cdepth_hard(ext, useData) (Captioning_models/Depth_caption_model/depth_train.py:27-325, 338-643) and
temp_anneal (:329-336).

The reference's COCO / "original" data pipelines (torchvision CocoCaptions + collate, util.py:52-221), its DPT
depth front-end and its vocabulary pickle are out of scope of this build and not present
offline, so `useData` additionally accepts "synthetic": procedurally generated RGB-D batches of the reference's
shapes.  The loop itself keeps the reference's structure: per-iteration fused train step (= depth_train.py:168-221),
per-epoch validation in eval mode (:238-303), loss CSVs (:232-233,302-303) and best-validation checkpoints of the
three state_dicts with the reference's file names (:306-322)."""
from __future__ import annotations

import os
from collections import deque

import numpy as np
import torch

from ... import native, synthetic as syn
from ..._lib import DicError
from ...engine import CaptionTrainer
from ..config import ConfigTrain

lam = 0.7             # depth_train.py:25
tqdm_disable = True   # depth_train.py:24


def temp_anneal(epoch_num):
    """temp = max(cos(pi * epoch / 360), 0.5) as a float32 tensor (depth_train.py:329-336)."""
    temp = np.array(np.cos(np.pi * (epoch_num / 360)))
    if temp <= 0.5:
        temp = np.array(0.5)
    return torch.from_numpy(temp.astype(np.float32)).clone()


def _synthetic_batches(config, rank: int, n: int, seed0: int, raw: bool = False):
    """raw=True: un-normalised RGB in [0,1) and no depth map (the DPT front-end predicts it, config 5)."""
    for i in range(n):
        s = seed0 + 7919 * i + rank
        imgs = syn.raw_images(config.batch_size, seed=s) if raw else syn.rgb_images(config.batch_size, seed=s)
        depth = None if raw else syn.depth_maps(config.batch_size, seed=s)
        caps, lens = syn.captions_fixed(config.batch_size, config.vocab_size, config.seq_len, seed=s)
        yield imgs, depth, caps, lens


class _DptFrontEnd:
    """Epoch-0 depth prediction + device-resident cache (depth_train.py:184-202): in epoch 0 the frozen DPT-Hybrid
    estimator predicts the depth map of every image, which is standardised, resized to 224x224 and stored under its key
    (the reference keys a CPU dict by the joined caption strings); later epochs fetch whole batches from the cache."""

    def __init__(self, config, capacity: int, share: "_DptFrontEnd" = None):
        from .DPT_model import DPT_Depthestimator
        from ..util import DepthCache
        # (the validation front-end shares the frozen estimator and owns the second cache, depth_dic_val: depth_train.py:258-275)
        self.dpt = share.dpt if share is not None else DPT_Depthestimator(getattr(config, "dpt_config", None)).to(config.device)   # :121,126-127 (frozen, eval)
        self.cache = DepthCache(capacity, device=config.device)
        self.forwards = self.hits = 0

    def depth_maps(self, epoch: int, keys, raw_gpu: torch.Tensor):
        from .. import util
        imgs, imgs_for_dep = util.device_transforms(raw_gpu)                                        # util.py:100-101
        if epoch == 0:
            depth = self.dpt.depth_maps_for_training(imgs_for_dep)                                  # :185-190
            self.cache.put(keys, depth)                                                             # :191-194
            self.forwards += 1
        else:
            depth = self.cache.get(keys)                                                            # :196-202
            self.hits += 1
        return imgs, depth


def _gumbel_draws(tmax: int, batch: int, epoch: int, iteration: int, rank: int, run: int) -> torch.Tensor:
    """u ~ U(0,1) [T,B,196]: an independent stream per (run, epoch, iteration, rank)."""
    seed = ((run * 1009 + epoch) * 1000003 + iteration) * 64 + rank
    return syn.gumbel_uniforms(tmax, batch, seed=seed)


def _train(ext, useData, hard: bool, config=None, process_group=None, stats=None, depth_branch: bool = True,
           tag: str = None, save_directory: str = None):
    """Shared loop of train_Cdepth_{soft,hard} and (depth_branch=False: no depth encoder, decoder-only optimiser, two
    checkpoints) of train_base_{soft,hard} (Base_caption_model/base_train.py:24-234, 248-460)."""
    config = config or ConfigTrain()
    if useData != "synthetic":
        raise DicError(f"useData={useData!r}: the MSCOCO / original-dataset loaders and the vocabulary pickle are outside "
                       "this build's scope (SURVEY.md 8f) and not available offline; use useData='synthetic' "
                       "(config.use_dpt = True adds the DPT depth front-end of BASELINE config 5)")
    if save_directory is None:
        save_directory = config.save_directory_Cdep_hard if hard else config.save_directory_Cdep_soft
    os.makedirs(save_directory, exist_ok=True)
    tag = tag or ("depth_hard" if hard else "depth_soft")
    train_loss_file = f"{save_directory}/{tag}_train_loss_{useData}{ext}.csv"
    val_loss_file = f"{save_directory}/{tag}_val_loss_{useData}{ext}.csv"
    rank = torch.distributed.get_rank(process_group) if process_group is not None else 0
    trainer = CaptionTrainer(config.vocab_size, device=config.device, seed=123 + int(ext), lr=config.lr, hard=hard,
                             dropout=config.dropout, lam=lam, process_group=process_group, use_depth=depth_branch,
                             conv_mode=getattr(config, "conv_mode", None))
    if rank == 0:
        print(f"[{tag}] frozen ResNet-152 convolutions in {trainer.conv_mode} arithmetic (config.conv_mode; bench.py --conv-mode "
              f"default: {native.DEFAULT_CONV_MODE})", flush=True)
    if stats is not None:
        stats.update(conv_mode=trainer.conv_mode)
    dev = config.device
    temp = torch.tensor(1.0)
    val_loss_best = float("inf")
    history = []
    use_dpt = depth_branch and bool(getattr(config, "use_dpt", False))        # BASELINE config 5: depth maps from the DPT front-end
    n_val = max(1, config.iters_per_epoch // 4)
    front = _DptFrontEnd(config, config.iters_per_epoch * config.batch_size) if use_dpt else None
    front_val = _DptFrontEnd(config, n_val * config.batch_size, share=front) if use_dpt else None
    for epoch in range(config.num_epochs):
        if hard and epoch % config.temp_sch == 0:                       # depth_train.py:481-483
            temp = temp_anneal(epoch)
        window, losses = deque(), []
        # with the DPT front-end the "dataset" is the same every epoch (that is what makes the depth cache meaningful)
        def on_device():
            """Batches of the epoch on the device (with the DPT front-end: normalised + depth maps from prediction / cache)."""
            for it, (imgs, depth, caps, lens) in enumerate(_synthetic_batches(config, rank, config.iters_per_epoch,
                                                                              0 if use_dpt else 1000 * epoch, raw=use_dpt)):
                if use_dpt:
                    imgs, depth = front.depth_maps(epoch, [f"{it}:{i}" for i in range(len(lens))], imgs.to(dev))
                yield it, imgs.to(dev), (depth.to(dev) if depth_branch else None), caps.to(dev), lens
        # two batches of look-ahead: the frozen RGB encoder runs ahead of the step on side streams (engine.prefetch_features)
        ahead, stream_it = [], on_device()
        for nxt in stream_it:
            ahead.append(nxt)
            if len(ahead) > 2:
                break
        while ahead:
            it, imgs, depth, caps, lens = ahead.pop(0)
            nxt = next(stream_it, None)
            if nxt is not None:
                ahead.append(nxt)
            # the reference draws a fresh torch.rand(bs, 196) per decode step of every iteration (attention.py:17); here
            # the T draws of one iteration come as one [T,B,196] tensor from a stream keyed by (epoch, iteration, rank)
            u = _gumbel_draws(max(lens) - 1, len(lens), epoch, it, rank, int(ext)).to(dev) if hard else None
            loss = trainer.train_step(imgs, depth, caps, lens, gumbel_u=u, temp=float(temp),
                                      next_imgs=[a[1] for a in ahead[:2]])
            losses.append(loss)                                         # device tensors: no per-iteration host sync
            window.append(loss)
            if len(window) > config.moving_avg:
                window.popleft()
        trainer.check_status()          # f16x2 overflow guard: raises DicError if a step of this epoch tripped it (engine.py)
        train_loss = float(torch.stack(losses).mean().item())
        if rank == 0:
            with open(train_loss_file, "a") as f:
                print(f"{epoch}, {train_loss}", file=f)
        # validation: eval-mode encoders, dropout off; soft = CE + regulariser (depth_train.py:248-292), hard =
        # decoder.eval_forward (Gumbel-max one-hot attention) with CE only (depth_train.py:555-610)
        # With the DPT front-end the validation images go through it too: predicted in epoch 0, read from the validation cache
        # (the reference's depth_dic_val) afterwards (depth_train.py:258-275) - same distribution as the training depth maps.
        val_losses = []
        for it, (imgs, depth, caps, lens) in enumerate(_synthetic_batches(config, rank, n_val, 777, raw=use_dpt)):
            if use_dpt:
                imgs, depth = front_val.depth_maps(epoch, [f"v{it}:{i}" for i in range(len(lens))], imgs.to(dev))
            u = _gumbel_draws(max(lens) - 1, len(lens), epoch, 100000 + it, rank, int(ext)).to(dev) if hard else None
            val_losses.append(trainer.eval_loss(imgs.to(dev), depth.to(dev) if depth_branch else None, caps.to(dev), lens,
                                                gumbel_u=u))
        val_loss = float(torch.stack(val_losses).mean().item())
        history.append((train_loss, val_loss))
        if rank == 0:
            with open(val_loss_file, "a") as f:
                print(f"{epoch}, {val_loss}", file=f)
            if val_loss < val_loss_best:                                # depth_train.py:306-322
                val_loss_best = val_loss
                sd = trainer.state_dicts()
                torch.save(sd["encoder"], f"{save_directory}/{tag}_encoder_best_{useData}{ext}.pth")
                torch.save(sd["decoder"], f"{save_directory}/{tag}_decoder_best_{useData}{ext}.pth")
                if depth_branch:
                    torch.save(sd["depth_encoder"], f"{save_directory}/{tag}_D_encoder_best_{useData}{ext}.pth")
    if stats is not None:
        stats.update(prefetch_dropped=trainer.prefetch_dropped)
    if stats is not None and front is not None:
        stats.update(dpt_forwards=front.forwards, cache_hits=front.hits, cache_entries=len(front.cache.slot),
                     val_dpt_forwards=front_val.forwards, val_cache_hits=front_val.hits,
                     val_cache_entries=len(front_val.cache.slot))
    return history


def train_Cdepth_soft(ext, useData, config=None, process_group=None, stats=None):
    return _train(ext, useData, hard=False, config=config, process_group=process_group, stats=stats)


def train_Cdepth_hard(ext, useData, config=None, process_group=None, stats=None):
    return _train(ext, useData, hard=True, config=config, process_group=process_group, stats=stats)
