import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_image_captioning_pub_amd import _lib
from depth_image_captioning_pub_amd._lib import ptr, stream_ptr, check
lib = _lib.load(); DEV = "cuda:0"
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
M = N = K = 4096
A = torch.randn(M, K, device=DEV); B = torch.randn(N, K, device=DEV); Cm = torch.empty(M, N, device=DEV)
for tile in (64, 128):
    for abl, name in ((0, "full"), (3, "no global loads (LDS stores kept)"), (1, "no global loads, no LDS stores"), (2, "MFMA only")):
        ft = abl * 1000 + tile
        def f(): check(lib.dic_gemm_f32(M, N, K, ptr(A), C.c_longlong(K), 0, ptr(B), C.c_longlong(K), 0, ptr(Cm), C.c_longlong(N), None, 0, 0, 1, None, C.c_size_t(0), ft, stream_ptr()))
        us = timeit(f)
        print(f"tile {tile:3d} {name:36s}: {us:8.1f} us  {2*M*N*K/us/1e6:6.1f} TF", flush=True)
