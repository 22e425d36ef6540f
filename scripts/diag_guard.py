"""Which kernel raises the f16x2 overflow guard word (bits: csrc/common.h) for a healthy forward?  python scripts/diag_guard.py [batch]"""
import sys
import torch
sys.path.insert(0, ".")
from depth_image_captioning_pub_amd import native, synthetic as syn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for layers in ((1, 1, 1, 1), (3, 8, 36, 3)):
    w = {k: v.cuda() for k, v in syn.resnet152_weights(seed=125, layers=layers).items()}
    rn = native.ResNetRunner(w, layers, conv_mode="f16x2")
    for seed in (123, 124):
        for train in (True, False):
            x = syn.rgb_images(B, seed=seed).cuda()
            y = rn.forward(x, train_bn=train, compact=True)
            torch.cuda.synchronize()
            print(layers, "batch", B, "seed", seed, "train" if train else "eval", "status", int(rn.status_word().item()),
                  "finite", bool(torch.isfinite(y).all()), "max", float(y.abs().max()) if torch.isfinite(y).all() else None, flush=True)
