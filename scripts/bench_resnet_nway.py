"""N frozen ResNet-152 forwards (batch 64 each, own workspace) on N HIP streams at once, for several persistent-grid sizes
(dic_conv_persistent_grid): wall time of one round of N concurrent forwards vs N x one forward alone.  Tells whether
convolutions of different forwards really run side by side when each launch asks for few CUs.
usage: python scripts/bench_resnet_nway.py [batch] [N,...] [G,...] [graph|eager]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_image_captioning_pub_amd import native, synthetic as syn, _lib

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NS = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,7").split(",")]
GS = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "224,98,49").split(",")]
MODE = sys.argv[4] if len(sys.argv) > 4 else "graph"
lib = _lib.load()
rn = syn.resnet152_weights(seed=125)
NMAX = max(NS)
base = {k: v.to(dev) for k, v in rn.items()}
stat = lambda k: k.endswith("running_mean") or k.endswith("running_var")
runners = [native.ResNetRunner({k: (v.clone() if stat(k) else v) for k, v in base.items()}, conv_mode="bf16x3") for _ in range(NMAX)]
imgs = [syn.rgb_images(B, seed=123 + i).to(dev) for i in range(NMAX)]
outs = [torch.empty((B, 49, 2048), device=dev) for _ in range(NMAX)]
streams = [torch.cuda.Stream() for _ in range(NMAX)]
for G in GS:
    _lib.check(lib.dic_conv_persistent_grid(G))
    graphs = []
    for i in range(NMAX):
        with torch.cuda.stream(streams[i]):
            runners[i].forward(imgs[i], True, out=outs[i], compact=True)
            streams[i].synchronize()
            if MODE == "graph":
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=streams[i], capture_error_mode="thread_local"):
                    runners[i].forward(imgs[i], True, out=outs[i], compact=True)
                graphs.append(g)
    torch.cuda.synchronize()

    def launch(i):
        with torch.cuda.stream(streams[i]):
            if MODE == "graph":
                graphs[i].replay()
            else:
                runners[i].forward(imgs[i], True, out=outs[i], compact=True)

    def run(n, iters=6):
        for i in range(n):
            launch(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            for i in range(n):
                launch(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    one = run(1)
    for n in NS:
        t = run(n)
        print(f"grid {G:4d} ({MODE}): {n} forwards at once {t:8.2f} ms = {t / n:6.2f} ms per forward (one alone {one:.2f} ms; "
              f"speed-up over serial {n * one / t:.2f}x)", flush=True)
