"""Greedy-decode evaluation loop with the reference's entry point Cdepth_evaluation(atten, useData)
(depth_evaluation.py:26-193): for every trained parameter triple, load the three state_dicts into the drop-in modules,
switch everything to eval mode, and for each validation batch run  dpt -> standardize -> Resize(224) -> depth_encoder,
encoder, decoder.batch_sample  (depth_evaluation.py:146-165), then turn the token ids into captions up to '<end>'
(:167-176).  Every tensor operation runs in libdic_hip.so; the decode keeps the previous token on the device (the reference
copies it to the host every step, depth_models.py:298-299).

Out of scope here as in SURVEY.md section 2: the COCO dataset / annotation files and the pycocoevalcap scorers
(BLEU / METEOR (Java) / ROUGE / CIDEr, evaluate_metrix.py) - `useData` must be "synthetic" (procedural images, a procedural
vocabulary); no scores are computed, the hypotheses are returned and written next to the checkpoints."""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from . import synthetic as syn
from ._lib import DicError
from .Captioning_models import util
from .Captioning_models.Base_caption_model.base_caption_models import CNNEncoder_Atten
from .Captioning_models.config import ConfigTrain
from .Captioning_models.Depth_caption_model.depth_models import (CD_RNNDecoderWithHardAttention,
                                                                 CD_RNNDecoderWithSoftAttention, Depth_CNN_endoder)
from .Captioning_models.Depth_caption_model.DPT_model import DPT_Depthestimator


def synthetic_vocabulary(vocab_size: int):
    """word <-> id dictionaries with the notebook's layout: ordinary words first, then <start>, <end>, <unk>, <null>
    (dataset/vocabulary_dict.ipynb cell 1)."""
    words = [f"w{i}" for i in range(vocab_size - 4)] + ["<start>", "<end>", "<unk>", "<null>"]
    return {w: i for i, w in enumerate(words)}, dict(enumerate(words))


def ids_to_captions(hypos_id: np.ndarray, id_to_word: Dict[int, str]) -> List[str]:
    """depth_evaluation.py:167-176: words up to (not including) the first '<end>'."""
    out = []
    for ids in hypos_id:
        line = []
        for i in ids:
            w = id_to_word[int(i)]
            if w == "<end>":
                break
            line.append(w)
        out.append(" ".join(line))
    return out


@torch.no_grad()
def Cdepth_evaluation(atten: str, useData: str, config=None, param_files: Optional[Dict[str, List[str]]] = None,
                      n_batches: int = 2, dpt: Optional[DPT_Depthestimator] = None):
    """Returns {key: {"hypotheses": [...], "ids": np.int64 [N,30]}} per parameter triple.  `param_files` maps a key to
    [encoder, decoder, depth-encoder] checkpoint file names inside the run's save directory (config.depth_*_parameter_files
    in the reference, config.py:131-136); default = the best-validation files train_Cdepth_* wrote for run 0."""
    if useData != "synthetic":
        raise DicError(f"useData={useData!r}: MSCOCO and the original dataset are not available offline; use 'synthetic'")
    if atten not in ("soft", "hard"):
        raise DicError("atten must be 'soft' or 'hard'")
    config = config or ConfigTrain()
    dev = config.device
    tag = f"depth_{atten}"
    save_directory = config.save_directory_Cdep_soft if atten == "soft" else config.save_directory_Cdep_hard
    if param_files is None:
        param_files = {"run0": [f"{tag}_encoder_best_synthetic0.pth", f"{tag}_decoder_best_synthetic0.pth",
                                f"{tag}_D_encoder_best_synthetic0.pth"]}
    word_to_id, id_to_word = synthetic_vocabulary(config.vocab_size)
    encoder = CNNEncoder_Atten(config.enc_img_size)                                          # depth_evaluation.py:108-129
    if atten == "soft":
        decoder = CD_RNNDecoderWithSoftAttention(config.dim_attention, config.dim_embedding, config.dim_encoder,
                                                 config.dim_hidden, config.vocab_size)
    else:
        decoder = CD_RNNDecoderWithHardAttention(config.dim_attention, config.dim_embedding, config.dim_encoder,
                                                 config.dim_hidden, config.vocab_size, dev, config.dropout)
    depth_encoder = Depth_CNN_endoder(config.enc_img_size)
    dpt = dpt if dpt is not None else DPT_Depthestimator(getattr(config, "dpt_config", None))
    for m in (encoder, decoder, depth_encoder, dpt):
        m.to(dev)
        m.eval()                                                                            # :131-134
    results = {}
    for key, (f_enc, f_dec, f_denc) in param_files.items():
        encoder.load_state_dict(torch.load(f"{save_directory}/{f_enc}", weights_only=True))       # :139-144
        decoder.load_state_dict(torch.load(f"{save_directory}/{f_dec}", weights_only=True))
        depth_encoder.load_state_dict(torch.load(f"{save_directory}/{f_denc}", weights_only=True))
        hypos_id = []
        for b in range(n_batches):
            raw = syn.raw_images(config.batch_size, seed=5000 + b).to(dev)
            imgs, imgs_for_dep = util.device_transforms(raw)
            depth_maps = dpt.depth_maps_for_training(imgs_for_dep)                          # :155-159
            depth_features = depth_encoder(depth_maps)                                      # :161
            feature = encoder(imgs)                                                         # :164
            hypos_id.append(decoder.batch_sample(feature, depth_features, word_to_id))     # :165
        hypos_id = np.concatenate(hypos_id)
        hypos_word = ids_to_captions(hypos_id, id_to_word)
        results[key] = {"hypotheses": hypos_word, "ids": hypos_id}
        # depth_evaluation.py:178-184 scores the hypotheses with BLEU / METEOR / CIDEr (pycocoevalcap + Java): out of scope
        #  - the hypotheses are the product of this path.
    with open(os.path.join(save_directory, f"{useData}_hypotheses.json"), "w") as f:
        json.dump({k: v["hypotheses"] for k, v in results.items()}, f)
    return results
