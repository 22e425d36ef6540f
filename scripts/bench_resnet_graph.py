"""ResNet-152 forward (batch 64, bf16x3): eager launches vs one captured hipGraph replay."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_image_captioning_pub_amd import native, synthetic as syn
DEV = "cuda:0"
rn = {k: v.to(DEV) for k, v in syn.resnet152_weights(seed=125).items()}
runner = native.ResNetRunner(rn, conv_mode="bf16x3")
imgs = syn.rgb_images(64, seed=123).to(DEV)
out = torch.empty((64, 196, 2048), device=DEV)
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(iters): fn()
    e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, (t1 - t0) / iters * 1e3
eager = lambda: runner.forward(imgs, train_bn=True, out=out)
g_ms, cpu_ms = timeit(eager)
print(f"eager : {g_ms:.3f} ms GPU per forward, {cpu_ms:.3f} ms CPU enqueue", flush=True)
ref = out.clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
graph = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    runner.forward(imgs, train_bn=True, out=out)          # warm-up on the capture stream
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=s):
        runner.forward(imgs, train_bn=True, out=out)
torch.cuda.synchronize()
out.zero_()
g2, c2 = timeit(lambda: graph.replay())
print(f"graph : {g2:.3f} ms GPU per forward, {c2:.3f} ms CPU enqueue", flush=True)
print("same output:", bool(torch.allclose(out, ref, rtol=1e-4, atol=1e-5)), float((out - ref).abs().max()))
