"""Two frozen ResNet-152 forwards (batch 64 each, own workspace / BatchNorm buffers) on two HIP streams concurrently vs
back to back: how much of a forward's dependent-launch gaps and shallow-grid idle time can a second forward fill?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_image_captioning_pub_amd import native, synthetic as syn

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rn = syn.resnet152_weights(seed=125)
runners = [native.ResNetRunner({k: v.to(dev).clone() for k, v in rn.items()}, conv_mode="bf16x3") for _ in range(2)]
imgs = [syn.rgb_images(B, seed=123 + i).to(dev) for i in range(2)]
outs = [torch.empty((B, 49, 2048), device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
graphs = []
for i in range(2):
    with torch.cuda.stream(streams[i]):
        runners[i].forward(imgs[i], True, out=outs[i], compact=True)
        streams[i].synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=streams[i], capture_error_mode="thread_local"):
            runners[i].forward(imgs[i], True, out=outs[i], compact=True)
        graphs.append(g)
torch.cuda.synchronize()


def run(concurrent, iters=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        if concurrent:
            for i in range(2):
                with torch.cuda.stream(streams[i]):
                    graphs[i].replay()
        else:
            with torch.cuda.stream(streams[0]):
                graphs[0].replay()
                graphs[0].replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


for _ in range(2):
    print(f"batch {B}: two forwards back to back {run(False):.2f} ms, concurrently on two streams {run(True):.2f} ms", flush=True)
