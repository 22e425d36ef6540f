"""How much does a concurrent stream of small dependent kernels slow the ResNet forward?  (one MI355X)
Cases: ResNet alone | + N tiny kernels on another stream (1-element add: pure launch/boundary cost) |
+ N medium bandwidth kernels (add on 100 MB) on another stream."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_image_captioning_pub_amd import native, synthetic as syn
DEV = "cuda:0"
rn = {k: v.to(DEV) for k, v in syn.resnet152_weights(seed=125).items()}
runner = native.ResNetRunner(rn, conv_mode="bf16x3")
imgs = syn.rgb_images(64, seed=123).to(DEV)
out = torch.empty((64, 196, 2048), device=DEV)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(sA):
    runner.forward(imgs, train_bn=True, out=out)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=sA):
    runner.forward(imgs, train_bn=True, out=out)
torch.cuda.synchronize()
tiny = torch.zeros(1, device=DEV)
big = torch.zeros(25_000_000, device=DEV)       # 100 MB
def run(n_tiny=0, n_big=0, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(sA):
            e0.record(); g.replay(); e1.record()
        with torch.cuda.stream(sB):
            b0.record()
            for _ in range(n_tiny): tiny.add_(1.0)
            for _ in range(n_big): big.add_(1.0)
            b1.record()
        torch.cuda.synchronize()
        ts.append((e0.elapsed_time(e1), b0.elapsed_time(b1)))
    ts.sort(); return ts[len(ts) // 2]
print("ResNet alone                    : %.2f ms" % run()[0])
for n in (100, 340, 1000):
    a, b = run(n_tiny=n)
    print(f"+ {n:4d} tiny kernels on stream B : ResNet {a:.2f} ms, stream B {b:.2f} ms")
for n in (10, 40):
    a, b = run(n_big=n)
    print(f"+ {n:4d} 100-MB adds on stream B  : ResNet {a:.2f} ms, stream B {b:.2f} ms")
torch.cuda.synchronize()
with torch.cuda.stream(sB):
    b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b0.record()
    for _ in range(340): tiny.add_(1.0)
    b1.record()
torch.cuda.synchronize(); print("340 tiny kernels alone: %.2f ms" % b0.elapsed_time(b1))
with torch.cuda.stream(sB):
    b0.record()
    for _ in range(40): big.add_(1.0)
    b1.record()
torch.cuda.synchronize(); print("40 big adds alone: %.2f ms" % b0.elapsed_time(b1))
