"""Micro-benchmark of the contraction kernel on GEMM and ResNet-layer shapes (B=64)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_image_captioning_pub_amd import _lib
from depth_image_captioning_pub_amd._lib import ptr, stream_ptr, check
lib = _lib.load()
DEV = "cuda:0"
TAIL = torch.empty(256*64*64, device="cuda:0") if "--tail" in __import__("sys").argv else None
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us
def gemm(M, N, K, tile):
    A = torch.randn(M, K, device=DEV); B = torch.randn(N, K, device=DEV); Cm = torch.empty(M, N, device=DEV)
    def f(): check(lib.dic_gemm_f32(M, N, K, ptr(A), C.c_longlong(K), 0, ptr(B), C.c_longlong(K), 0, ptr(Cm), C.c_longlong(N), None, 0, 0, 1, None, C.c_size_t(0), tile, stream_ptr()))
    us = timeit(f)
    print(f"gemm  {M:7d}x{N:5d}x{K:5d} tile {tile:3d}: {us:8.1f} us  {2*M*N*K/us/1e6:6.1f} TF", flush=True)
def conv(Bn, H, Cc, CO, k, s, p, tile, stats=True):
    x = torch.randn(Bn, H, H, Cc, device=DEV); w = torch.randn(CO, k, k, Cc, device=DEV)
    OH = (H + 2*p - k)//s + 1
    y = torch.empty(Bn, OH, OH, CO, device=DEV)
    M = Bn*OH*OH
    part = torch.empty((M//64+2)*2*CO, device=DEV)
    mt = C.c_int(0)
    def f(): check(lib.dic_conv2d_fwd(ptr(x), Bn, H, H, Cc, 0, ptr(w), None, CO, k, k, s, p, ptr(y), ptr(part) if stats else None, C.byref(mt), tile, ptr(TAIL), stream_ptr()))
    us = timeit(f)
    fl = 2*M*CO*k*k*Cc
    print(f"conv  M={M:7d} N={CO:5d} K={k*k*Cc:5d} tile {tile:3d} stats={int(stats)}: {us:8.1f} us  {fl/us/1e6:6.1f} TF  tiles={(-(-M//tile))*(-(-CO//tile))}", flush=True)
import sys as _s
if len(_s.argv) > 1 and _s.argv[1].isdigit(): lib.dic_debug_force_staged_gemm(int(_s.argv[1])); print("force =", _s.argv[1])
gemm(4096, 4096, 4096, 128); gemm(4096, 4096, 4096, 64)
gemm(8192, 8192, 1024, 128)
for tile in (64, 128):
    conv(64, 14, 1024, 256, 1, 1, 0, tile)      # l3 c1
    conv(64, 14, 256, 256, 3, 1, 1, tile)       # l3 c2
    conv(64, 14, 256, 1024, 1, 1, 0, tile)      # l3 c3
conv(64, 14, 256, 256, 3, 1, 1, 64, stats=False)
conv(256, 14, 256, 256, 3, 1, 1, 64)            # B=256: tail-free-ish
conv(256, 14, 256, 256, 3, 1, 1, 128)
conv(64, 56, 64, 64, 3, 1, 1, 64); conv(64, 56, 64, 256, 1, 1, 0, 64); conv(64, 56, 64, 256, 1, 1, 0, 128)
conv(64, 28, 128, 128, 3, 1, 1, 64); conv(64, 28, 128, 128, 3, 1, 1, 128)
