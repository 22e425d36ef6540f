"""Stage times of the main stream (depth encoder, decoder, backward, AdamW) measured with events INSIDE the pipelined step,
i.e. while two ResNet forwards run on the side streams, against the same stages alone on the chip."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_image_captioning_pub_amd import synthetic as syn, _lib
from depth_image_captioning_pub_amd.engine import CaptionTrainer
for c in sys.argv[1:]:
    _lib.load().dic_debug_force_staged_gemm(int(c))
dev, B, V, T = "cuda:0", 64, 10000, 20
tr = CaptionTrainer(V, device=dev, seed=123, conv_mode="bf16x3")
imgs = syn.rgb_images(B, seed=123).to(dev); depth = syn.depth_maps(B, seed=123).to(dev)
caps, lens = syn.captions_fixed(B, V, T, seed=123); caps = caps.to(dev)
pipe = {"next_imgs": [imgs, imgs]}
for _ in range(8): tr.train_step(imgs, depth, caps, lens, **pipe)
torch.cuda.synchronize()
tr.timing = True
acc = {}
N = 10
t0 = time.perf_counter()
for _ in range(N):
    tr.train_step(imgs, depth, caps, lens, **pipe)
    torch.cuda.synchronize()
    for k, v in tr.stage_ms().items(): acc[k] = acc.get(k, 0.0) + v / N
print("pipelined (two forwards in flight), per step incl. a sync per step:", {k: round(v, 3) for k, v in acc.items()}, "sum", round(sum(acc.values()), 3),
      f"wall {(time.perf_counter() - t0) / N * 1e3:.2f} ms")
tr.prefetched = None
feats = tr.resnet.forward(imgs, True, compact=tr.compact_ok).clone(); torch.cuda.synchronize()
acc = {}
for _ in range(N):
    tr.train_step(None, depth, caps, lens, precomputed_features=feats)
    torch.cuda.synchronize()
    for k, v in tr.stage_ms().items(): acc[k] = acc.get(k, 0.0) + v / N
print("alone on the chip:", {k: round(v, 3) for k, v in acc.items()}, "sum", round(sum(acc.values()), 3))
