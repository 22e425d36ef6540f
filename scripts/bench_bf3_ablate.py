import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_image_captioning_pub_amd import _lib
from depth_image_captioning_pub_amd._lib import ptr, stream_ptr, check
lib = _lib.load(); DEV = "cuda:0"
def split(x):
    n = x.numel(); hi = torch.empty(n, dtype=torch.int16, device=DEV); mid = torch.empty_like(hi); lo = torch.empty_like(hi)
    check(lib.dic_split_bf16x3(ptr(x), C.c_longlong(n), ptr(hi), ptr(mid), ptr(lo), stream_ptr())); return hi, mid, lo
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
lib.dic_debug_force_staged_gemm(11)
for (M, N, K) in [(4096, 4096, 4096), (12544, 256, 2304), (50176, 256, 2304), (12544, 1024, 256)]:
    A = torch.randn(M, K, device=DEV); B = torch.randn(N, K, device=DEV); a = split(A); b = split(B); Cb = torch.empty(M, N, device=DEV)
    out = []
    for abl, name in ((0, "full"), (1, "no DMA"), (2, "MFMA only")):
        lib.dic_debug_force_staged_gemm(50 + abl)
        f = lambda: check(lib.dic_gemm_bf16x3(M, N, K, ptr(a[0]), ptr(a[1]), ptr(a[2]), C.c_longlong(K), ptr(b[0]), ptr(b[1]), ptr(b[2]), C.c_longlong(K), ptr(Cb), C.c_longlong(N), None, stream_ptr()))
        us = timeit(f); out.append(f"{name}: {us:7.1f}us {2*M*N*K/us/1e6:6.1f}TF")
    print(f"M={M:6d} N={N:5d} K={K:5d} | " + " | ".join(out), flush=True)
