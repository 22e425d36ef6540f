"""Deep-pipelined 128x128 bf16x3 kernel (debug code 23) against the 64x64 (11), 128x64 (21) and plain 128x128 (22) tiles:
bit-identity of the results (same summation order) and time per launch, on ResNet-152 layer shapes at batch 64 / 256 and on
square problems.  usage: bench_bf3_pipe.py [batch]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_image_captioning_pub_amd import _lib
from depth_image_captioning_pub_amd._lib import ptr, stream_ptr, check
lib = _lib.load(); DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
def split(x):
    R, K = x.shape; Rp = (R + 1) // 2 * 2
    out = [torch.empty(Rp * K, dtype=torch.int16, device=DEV) for _ in range(3)]
    check(lib.dic_split_bf16x3_paired(ptr(x), C.c_longlong(R), K, ptr(out[0]), ptr(out[1]), ptr(out[2]), stream_ptr())); return out
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
shapes = [(B * 196, 256, 1024), (B * 196, 1024, 256), (B * 784, 128, 512), (B * 784, 512, 128), (B * 49, 512, 2048), (B * 49, 2048, 512),
          (B * 3136, 256, 64), (4096, 4096, 4096), (8192, 8192, 4096), (1000, 300, 96), (129, 130, 64), (40000, 384, 160)]
for (M, N, K) in shapes:
    A = torch.randn(M, K, device=DEV); Bm = torch.randn(N, K, device=DEV); a = split(A); b = split(Bm)
    outs, res = {}, []
    for code in (11, 21, 24, 26):
        lib.dic_debug_force_staged_gemm(code)
        Cm = torch.full((M, N), float("nan"), device=DEV)
        f = lambda: check(lib.dic_gemm_bf16x3_paired(M, N, K, ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(b[0]), ptr(b[1]), ptr(b[2]), ptr(Cm), C.c_longlong(N), None, stream_ptr()))
        t = timeit(f); outs[code] = Cm
        res.append(f"{code}: {t:7.1f}us {2*M*N*K/t/1e6:6.1f}TF")
    lib.dic_debug_force_staged_gemm(20)
    same = all(torch.equal(outs[11], outs[c]) for c in (21, 24, 26))
    ref = A.double() @ Bm.double().t()
    err = float((outs[26].double() - ref).abs().max() / ref.abs().max())
    print(f"M={M:6d} N={N:5d} K={K:5d} | " + " | ".join(res) + f" | bit-identical {same} | rel err vs fp64 {err:.1e}", flush=True)
