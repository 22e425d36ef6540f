"""Is the pipelined step bound by the host?  Enqueue time of train_step (no synchronisation) against the measured step time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_image_captioning_pub_amd import synthetic as syn
from depth_image_captioning_pub_amd.engine import CaptionTrainer
dev, B, V, T = "cuda:0", 64, 10000, 20
tr = CaptionTrainer(V, device=dev, seed=123, conv_mode="bf16x3")
imgs = syn.rgb_images(B, seed=123).to(dev); depth = syn.depth_maps(B, seed=123).to(dev)
caps, lens = syn.captions_fixed(B, V, T, seed=123); caps = caps.to(dev)
pipe = {"next_imgs": [imgs, imgs]}
for _ in range(8): tr.train_step(imgs, depth, caps, lens, **pipe)
torch.cuda.synchronize()
N = 40
t0 = time.perf_counter(); host = []
for _ in range(N):
    a = time.perf_counter(); tr.train_step(imgs, depth, caps, lens, **pipe); host.append(time.perf_counter() - a)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
host.sort()
print(f"enqueue per step: mean {t_enq / N * 1e3:.2f} ms, median {host[N // 2] * 1e3:.2f} ms, max {host[-1] * 1e3:.2f} ms; step time (with final sync) {t_all / N * 1e3:.2f} ms")
# the same with the main-stream work only (features precomputed): how long does the host need for the non-ResNet part?
feats = tr.resnet.forward(imgs, True, compact=tr.compact_ok).clone(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N): tr.train_step(None, depth, caps, lens, precomputed_features=feats)
t_enq2 = time.perf_counter() - t0
torch.cuda.synchronize(); t_all2 = time.perf_counter() - t0
print(f"main-stream work only: enqueue {t_enq2 / N * 1e3:.2f} ms per step, GPU-complete {t_all2 / N * 1e3:.2f} ms per step")
