"""Persistent warp-specialised kernel (code 24): full vs no DMA inside the loop (code 51), per K tile, on a square problem and
on ResNet layer shapes at batch 256.  Says how much of a K tile is still spent waiting for operands."""
import sys, os, ctypes as C
os.environ.setdefault("DIC_LIB", "experiments")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_image_captioning_pub_amd import _lib
from depth_image_captioning_pub_amd._lib import ptr, stream_ptr, check
lib = _lib.load(); DEV = "cuda:0"
def split(x):
    R, K = x.shape
    out = [torch.empty((R + 1) // 2 * 2 * K, dtype=torch.int16, device=DEV) for _ in range(3)]
    check(lib.dic_split_bf16x3_paired(ptr(x), C.c_longlong(R), K, ptr(out[0]), ptr(out[1]), ptr(out[2]), stream_ptr())); return out
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
os.environ.setdefault("DIC_LIB", "experiments")      # (set before _lib.load above when run as a script: see the top of the file)
for (M, N, K) in [(4096, 4096, 4096), (50176, 256, 1024), (50176, 1024, 256), (12544, 256, 1024), (12544, 1024, 256)]:
    A = torch.randn(M, K, device=DEV); B = torch.randn(N, K, device=DEV); a = split(A); b = split(B)
    Cm = torch.empty(M, N, device=DEV)
    f = lambda: check(lib.dic_gemm_bf16x3_paired(M, N, K, ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(b[0]), ptr(b[1]), ptr(b[2]), ptr(Cm), C.c_longlong(N), None, stream_ptr()))
    lib.dic_debug_force_staged_gemm(24)
    T = -(-M // 128) * (N // 128); per_wg = -(-T // 224)
    slots = per_wg * (K // 32)
    for abl, name in ((0, "full"), (2, "2-stage ring"), (1, "no DMA in loop"), (4, "A cache-hot"), (5, "A and B cache-hot")):
        lib.dic_debug_force_staged_gemm(50 + abl)
        t = timeit(f)
        print(f"M={M:6d} N={N:5d} K={K:5d} {name:16s}: {t:8.1f} us = {t / slots * 2400:6.0f} nominal cycles per K tile of the busiest workgroup ({per_wg} tiles), {2*M*N*K*6/t/1e6/2500:.3f} of bf16 peak", flush=True)
    lib.dic_debug_force_staged_gemm(50); lib.dic_debug_force_staged_gemm(20)
