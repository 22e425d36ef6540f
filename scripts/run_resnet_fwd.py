"""One process, N ResNet-152 forwards (bf16x3, train-mode BN) at --batch; meant to run under
`rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 scripts/run_resnet_fwd.py`; scripts/trace_layers.py then
lists the last forward's dispatches in order (duration, grid, gap to the previous kernel)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from depth_image_captioning_pub_amd import native, synthetic as syn  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--iters", type=int, default=4)
ap.add_argument("--conv-mode", default="bf16x3")
ap.add_argument("--switches", default="", help="comma-separated dic_debug_force_staged_gemm codes")
a = ap.parse_args()
DEV = "cuda:0"
for code in filter(None, a.switches.split(",")):
    assert native._lib.load().dic_debug_force_staged_gemm(int(code)) == 0, code
rn = {k: v.to(DEV) for k, v in syn.resnet152_weights(seed=125).items()}
runner = native.ResNetRunner(rn, conv_mode=a.conv_mode)
imgs = syn.rgb_images(a.batch, seed=123).to(DEV)
out = torch.empty((a.batch, 49, 2048), device=DEV)
for _ in range(a.iters):
    runner.forward(imgs, train_bn=True, out=out, compact=True)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
runner.forward(imgs, train_bn=True, out=out, compact=True)
e1.record()
torch.cuda.synchronize()
print(f"last forward: {e0.elapsed_time(e1):.3f} ms at batch {a.batch}")
