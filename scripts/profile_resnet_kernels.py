"""Per-instantiation totals of the contraction kernels over one ResNet-152 forward (dic_profile_begin/end: HIP events around
every launch).  usage: profile_resnet_kernels.py [--batch N] [debug codes ...]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
BATCH = 64
if len(sys.argv) > 2 and sys.argv[1] == '--batch':
    BATCH = int(sys.argv[2]); del sys.argv[1:3]
from depth_image_captioning_pub_amd import native, synthetic as syn, _lib
DEV = "cuda:0"
lib = _lib.load()
for c in sys.argv[1:]:
    lib.dic_debug_force_staged_gemm(int(c))
rn = {k: v.to(DEV) for k, v in syn.resnet152_weights(seed=125).items()}
runner = native.ResNetRunner(rn, conv_mode="bf16x3")
imgs = syn.rgb_images(BATCH, seed=123).to(DEV)
for _ in range(2): runner.forward(imgs, train_bn=True)
torch.cuda.synchronize()
_lib.check(lib.dic_profile_begin(), "begin")
runner.forward(imgs, train_bn=True)
n = 64
keys = (C.c_int * n)(); ms = (C.c_double * n)(); fl = (C.c_double * n)(); cnt = (C.c_longlong * n)(); nout = C.c_int(0)
_lib.check(lib.dic_profile_end(n, keys, ms, fl, cnt, C.byref(nout)), "end")
KIND = {0: "rowk", 2: "im2col"}
TILE = {0: "64x64", 1: "64x128", 2: "128x64", 3: "128x128", 4: "128x128 pipe", 5: "128x128 persistent"}
tot = 0.0
for i in sorted(range(nout.value), key=lambda i: -ms[i]):
    k = keys[i]
    name = f"bf3 {KIND.get((k - 2000) // 10, '?')} {TILE.get((k - 2000) % 10, '?')}" if k >= 2000 else f"key {k}"
    tot += ms[i]
    print(f"{name:32s} launches {cnt[i]:4d}  total {ms[i]:7.3f} ms  avg {ms[i] / cnt[i] * 1e3:7.1f} us  {fl[i] / ms[i] / 1e9:6.1f} TF-eq  ({fl[i] * 6 / ms[i] / 1e9 / 2500:.3f} of bf16 peak)")
print(f"contractions total {tot:.3f} ms at batch {BATCH}")
