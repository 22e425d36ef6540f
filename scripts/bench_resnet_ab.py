"""Stable A/B of library debug switches on the ResNet-152 forward alone (bf16x3, train-mode BN).
usage: python scripts/bench_resnet_ab.py [--batch N] CODE_A CODE_B [...]   (codes for dic_debug_force_staged_gemm)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
BATCH = 64
if len(sys.argv) > 2 and sys.argv[1] == '--batch':
    BATCH = int(sys.argv[2]); del sys.argv[1:3]
from depth_image_captioning_pub_amd import native, synthetic as syn, _lib
DEV = "cuda:0"
lib = _lib.load()
rn = {k: v.to(DEV) for k, v in syn.resnet152_weights(seed=125).items()}
runner = native.ResNetRunner(rn, conv_mode="bf16x3")
imgs = syn.rgb_images(BATCH, seed=123).to(DEV)
out = torch.empty((BATCH, 196, 2048), device=DEV)
def timeit(iters=20):
    for _ in range(3): runner.forward(imgs, train_bn=True, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): runner.forward(imgs, train_bn=True, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
codes = [tuple(int(x) for x in c.split('+')) for c in sys.argv[1:]] or [(0,)]   # 'a+b' applies several switches
ref_out = None
for rep in range(4):
    res = []
    for c in codes:
        for x in c: lib.dic_debug_force_staged_gemm(x)
        t = timeit()
        if ref_out is None: ref_out = out.clone()
        diff = float((out - ref_out).abs().max() / ref_out.abs().max())
        res.append(f"code {'+'.join(map(str, c))}: {t:.3f} ms (rel.diff vs first {diff:.1e})")
    print(" | ".join(res), flush=True)
