"""Training harness with the reference's entry points train_base_soft(ext, useData) / train_base_hard(ext, useData)
(Captioning_models/Base_caption_model/base_train.py:24-234, 248-460): the Show-Attend-and-Tell captioner WITHOUT the depth
branch - frozen ResNet-152 (train() mode = batch-statistics BatchNorm, :139) -> RNNDecoderWith{Soft,Hard}Attention -> CE
(+ 0.7 x attention regulariser for soft, :160-162) -> AdamW over the decoder's parameters only (:115) - BASELINE config 1.

Same fused engine as the depth path with the depth branch switched off (CaptionTrainer(use_depth=False): dic_decoder_fwd /
_bwd with feat_depth = NULL).  Outputs keep the reference's names: base_{soft,hard}_{train,val}_loss_{useData}{ext}.csv,
base_{soft,hard}_{encoder,decoder}_best_{useData}{ext}.pth.  Reference quirk kept: train_base_hard also writes into
config.save_directory_soft (base_train.py:253,258).  As in depth_train.py the COCO loaders are out of scope: useData must be
"synthetic"."""
from __future__ import annotations

from ..config import ConfigTrain
from ..Depth_caption_model.depth_train import _train, temp_anneal      # noqa: F401  (temp_anneal: base_train.py:239-246)

lam = 0.7             # base_train.py:22
tqdm_disable = True   # base_train.py:21


def train_base_soft(ext, useData, config=None, process_group=None, stats=None):
    config = config or ConfigTrain()
    return _train(ext, useData, hard=False, config=config, process_group=process_group, stats=stats, depth_branch=False,
                  tag="base_soft", save_directory=config.save_directory_soft)


def train_base_hard(ext, useData, config=None, process_group=None, stats=None):
    config = config or ConfigTrain()
    return _train(ext, useData, hard=True, config=config, process_group=process_group, stats=stats, depth_branch=False,
                  tag="base_hard", save_directory=config.save_directory_soft)      # (sic: base_train.py:253)
