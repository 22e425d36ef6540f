"""CPU experiment (no GPU): would Winograd F(2x2,3x3) for the stride-1 3x3 convolutions of ResNet-152 survive the parity bar?
Evaluates the oracle's ResNet-152 (batch-statistics BatchNorm) three ways on the same inputs - fp64 direct, fp32 direct, fp32 with
the 3x3/stride-1 convolutions done by Winograd F(2x2,3x3) with fp32 transforms - and prints the feature error of the two fp32
evaluations against fp64 (the parity test demands HIP error <= 2x the fp32 oracle's)."""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from depth_image_captioning_pub_amd import synthetic as syn
from oracle import captioning_oracle as orc

G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def winograd_conv3x3(x, w):
    """x [B,C,H,W] (H, W even), w [K,C,3,3], padding 1, stride 1 -> [B,K,H,W]; every product / sum in x.dtype."""
    B, C, H, W = x.shape
    K = w.shape[0]
    g, bt, at = G.to(x.dtype), BT.to(x.dtype), AT.to(x.dtype)
    U = torch.einsum("ij,kcjl,ml->imkc", g, w, g)                       # [4,4,K,C]
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                          # [B,C,H/2,W/2,4,4]
    V = torch.einsum("ij,bcxyjl,ml->imbxyc", bt, tiles, bt)             # [4,4,B,H/2,W/2,C]
    M = torch.einsum("imbxyc,imkc->imbxyk", V, U)                       # 16 GEMMs
    Y = torch.einsum("pi,imbxyk,qm->bkxpyq", at, M, at)                 # [B,K,H/2,2,W/2,2]
    return Y.reshape(B, K, H, W)


def features(w, imgs, wino):
    with torch.no_grad():
        x = F.conv2d(imgs, w["backbone.0.weight"], None, stride=2, padding=3)
        x = torch.relu(orc.resnet_bn(x, w, "backbone.1.", True))
        x = F.max_pool2d(x, 3, stride=2, padding=1)
        for li, nblocks in enumerate(orc.RESNET152_LAYERS):
            for bi in range(nblocks):
                p = f"backbone.{4 + li}.{bi}."
                stride = 2 if (li > 0 and bi == 0) else 1
                idt = x
                y = torch.relu(orc.resnet_bn(F.conv2d(x, w[p + "conv1.weight"]), w, p + "bn1.", True))
                if wino and stride == 1 and y.shape[-1] % 2 == 0 and li in wino:
                    y = winograd_conv3x3(y, w[p + "conv2.weight"])
                else:
                    y = F.conv2d(y, w[p + "conv2.weight"], stride=stride, padding=1)
                y = torch.relu(orc.resnet_bn(y, w, p + "bn2.", True))
                y = orc.resnet_bn(F.conv2d(y, w[p + "conv3.weight"]), w, p + "bn3.", True)
                if bi == 0:
                    idt = orc.resnet_bn(F.conv2d(x, w[p + "downsample.0.weight"], stride=stride), w, p + "downsample.1.", True)
                x = torch.relu(y + idt)
        return x


B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.set_num_threads(8)
rn = syn.resnet152_weights(seed=125)
imgs = syn.rgb_images(B, seed=123)
f64 = features({k: v.double() for k, v in rn.items()}, imgs.double(), ())
s = float(f64.abs().max())
for name, wino in (("fp32 direct", ()), ("fp32 Winograd in layer 3 (14x14)", (2,)), ("fp32 Winograd in layers 1-3", (0, 1, 2))):
    f = features(copy.deepcopy(rn), imgs, wino)
    print(f"batch {B}: {name:36s} max |err| vs fp64 = {float((f.double() - f64).abs().max()) / s:.3e} of scale", flush=True)
# single-layer error of the convolution itself
x = torch.randn(4, 256, 14, 14); w = torch.randn(256, 256, 3, 3) / 48
ref = F.conv2d(x.double(), w.double(), padding=1)
for name, y in (("direct fp32", F.conv2d(x, w, padding=1)), ("Winograd fp32", winograd_conv3x3(x, w))):
    print(f"one 256->256 3x3 convolution, {name}: max err {float((y.double() - ref).abs().max()) / float(ref.abs().max()):.2e}")
