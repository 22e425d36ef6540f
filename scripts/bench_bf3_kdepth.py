"""bf16x3 kernel (paired layout) throughput vs K at a fixed tile count: how much do per-tile prologue/epilogue cost?"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_image_captioning_pub_amd import _lib
from depth_image_captioning_pub_amd._lib import ptr, stream_ptr, check
lib = _lib.load(); DEV = "cuda:0"
def split(x):
    R, K = x.shape; Rp = (R + 1) // 2 * 2
    out = [torch.empty(Rp * K, dtype=torch.int16, device=DEV) for _ in range(3)]
    check(lib.dic_split_bf16x3_paired(ptr(x), C.c_longlong(R), K, ptr(out[0]), ptr(out[1]), ptr(out[2]), stream_ptr())); return out
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, N) in [(12544, 1024), (12288, 1024), (12544, 256), (12288, 256), (50176, 512)]:
    for K in [64, 128, 256, 512, 1024, 2048, 4096]:
        A = torch.randn(M, K, device=DEV); B = torch.randn(N, K, device=DEV); a = split(A); b = split(B)
        Cm = torch.empty(M, N, device=DEV)
        f = lambda: check(lib.dic_gemm_bf16x3_paired(M, N, K, ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(b[0]), ptr(b[1]), ptr(b[2]), ptr(Cm), C.c_longlong(N), None, stream_ptr()))
        t = timeit(f)
        tiles = ((M + 63) // 64) * ((N + 63) // 64)
        print(f"M={M:6d} N={N:5d} K={K:5d} tiles={tiles:5d} ({tiles/768:5.2f} x768) {t:8.1f}us {2*M*N*K/t/1e6:6.1f}TF  us/ktile-round={t/(K/32)/max(1,tiles/768):6.3f}", flush=True)
