"""Hyper-parameters of the reference's ConfigTrain (Captioning_models/config.py:7-70) for the hot path,
plus the knobs a synthetic / multi-GPU run needs (vocab, batch, seq_len, world size)."""
from __future__ import annotations

import os


class ConfigTrain(object):
    def __init__(self):
        self.cwd = os.getcwd()
        self.enc_img_size = 14          # config.py:11
        self.dim_attention = 128        # :12
        self.dim_embedding = 128        # :13
        self.dim_encoder = 2048         # :14
        self.dim_hidden = 128           # :15
        self.lr = 0.001                 # :20
        self.dropout = 0.5              # :21
        self.batch_size = 30            # :22
        self.num_epochs = 150           # :23
        self.lr_drop = [20]             # :25 (scheduler is built but never stepped: quirk Q2)
        self.temp_sch = 10              # :26 temperature re-annealed every 10 epochs (hard path)
        self.device = "cuda:0"          # :68
        self.conv_mode = "f16x2"        # (not in the reference) arithmetic of the frozen ResNet-152: "f16x2" = the benchmarked mode
                                        # (native.DEFAULT_CONV_MODE; overflow-guarded), "bf16x3" / "fp32" = exact operands
        self.moving_avg = 100           # :71
        self.save_directory_soft = self.cwd + "/exp_result/base_soft"           # config.py:45 (base-soft; base_train.py:253 also puts base-hard here)
        self.save_directory_hard = self.cwd + "/exp_result/base_hard"           # :51
        self.save_directory_Cdep_soft = self.cwd + "/exp_result/CNN_depth_soft"
        self.save_directory_Cdep_hard = self.cwd + "/exp_result/CNN_depth_hard"
        # synthetic-run knobs (no dataset / vocabulary ships with the reference)
        self.vocab_size = 10000
        self.seq_len = 20
        self.iters_per_epoch = 20
        self.use_dpt = False            # True = BASELINE config 5: depth maps predicted by the DPT-Hybrid front-end in epoch 0
        self.dpt_config = None          # synthetic.DptConfig (None = vitb_rn50_384)
