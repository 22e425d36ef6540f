"""CPU experiment: would a two-term fp16 split (x = h1 + h2, three matrix-core products h1*h1' + h1*h2' + h2*h1') keep the ResNet-152
forward as close to an fp64 evaluation as the three-term bf16 split (six products) the library uses?  Emulates both inside the
oracle's forward (every convolution replaced by the sum of the split products, each an fp32 convolution) at B=2, 224x224, train-mode
BatchNorm, synthetic weights, and prints max / rms error of the final feature map against fp64.  python3 scripts/probe_fp16x2_numerics.py"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from depth_image_captioning_pub_amd import synthetic as syn  # noqa: E402
from oracle import captioning_oracle as orc  # noqa: E402

real_conv = F.conv2d


def split_bf16(x, n):
    out, r = [], x
    for _ in range(n):
        t = r.to(torch.bfloat16).to(torch.float32)
        out.append(t)
        r = r - t
    return out


def split_fp16(x, n):
    m = float(x.abs().max())
    s = 2.0 ** math.floor(14 - math.log2(m)) if m > 0 else 1.0          # largest element lands in (2^13, 2^14]
    out, r = [], x * s
    for _ in range(n):
        t = r.to(torch.float16).to(torch.float32)
        out.append(t)
        r = r - t
    return out, s


def make_conv(kind):
    def conv(x, w, b=None, **kw):
        if x.dtype != torch.float32:
            return real_conv(x, w, b, **kw)
        if kind == "fp32":
            return real_conv(x, w, b, **kw)
        if kind == "bf16x3":
            xs, ws = split_bf16(x, 3), split_bf16(w, 3)
            pairs = [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]
            y = sum(real_conv(xs[i], ws[j], None, **kw) for i, j in reversed(pairs))
            return y
        nprod = int(kind[-1])
        (xs, sx), (ws, sw) = split_fp16(x, 2), split_fp16(w, 2)
        pairs = [(0, 0), (0, 1), (1, 0), (1, 1)][:nprod]
        y = sum(real_conv(xs[i], ws[j], None, **kw) for i, j in reversed(pairs))
        return y * (1.0 / (sx * sw))
    return conv


w = syn.resnet152_weights(seed=125)
x = syn.rgb_images(2, seed=123)
orc.F.conv2d = real_conv
y64 = orc.resnet152_features({k: v.double() for k, v in w.items()}, x.double(), train_bn=True)
scale = float(y64.abs().max())
for kind in ("fp32", "bf16x3", "fp16x2_3", "fp16x2_4"):
    orc.F.conv2d = make_conv(kind)
    y = orc.resnet152_features({k: v.clone() for k, v in w.items()}, x, train_bn=True)
    d = (y.double() - y64)
    print(f"{kind:9s}: max |err| / max = {float(d.abs().max()) / scale:.3e}   rms err / rms = {float(d.pow(2).mean().sqrt() / y64.pow(2).mean().sqrt()):.3e}", flush=True)
orc.F.conv2d = real_conv
