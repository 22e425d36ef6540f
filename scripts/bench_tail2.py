import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_image_captioning_pub_amd import _lib
from depth_image_captioning_pub_amd._lib import ptr, stream_ptr, check
lib = _lib.load(); DEV = "cuda:0"
TAIL = torch.empty(256*64*64, device=DEV)
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
def conv(Bn, H, Cc, CO, k, tail):
    x = torch.randn(Bn, H, H, Cc, device=DEV); w = torch.randn(CO, k, k, Cc, device=DEV)
    y = torch.empty(Bn, H, H, CO, device=DEV); M = Bn*H*H
    part = torch.empty((M//64+2)*2*CO, device=DEV); mt = C.c_int(0)
    def f(): check(lib.dic_conv2d_fwd(ptr(x), Bn, H, H, Cc, 0, ptr(w), None, CO, k, k, 1, k//2, ptr(y), ptr(part), C.byref(mt), 64, ptr(tail), stream_ptr()))
    return timeit(f)
for name, mode in (("plain", None), ("tail+fix", 13), ("tail, fixup skipped", 12)):
    if mode: lib.dic_debug_force_staged_gemm(mode)
    us = conv(64, 14, 256, 256, 3, None if mode is None else TAIL)
    us1 = conv(64, 14, 1024, 256, 1, None if mode is None else TAIL)
    print(f"{name:22s}: c2 {us:7.1f} us   c1 {us1:7.1f} us", flush=True)
