import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_image_captioning_pub_amd import _lib
from depth_image_captioning_pub_amd._lib import ptr, stream_ptr, check
lib = _lib.load(); DEV = "cuda:0"
TAIL = torch.empty(256*64*64, device="cuda:0") if "--tail" in __import__("sys").argv else None
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
def conv(Bn, H, Cc, CO, k, s, p, tile=64):
    x = torch.randn(Bn, H, H, Cc, device=DEV); w = torch.randn(CO, k, k, Cc, device=DEV)
    OH = (H + 2*p - k)//s + 1; y = torch.empty(Bn, OH, OH, CO, device=DEV); M = Bn*OH*OH
    part = torch.empty((M//64+2)*2*CO, device=DEV); mt = C.c_int(0)
    def f(): check(lib.dic_conv2d_fwd(ptr(x), Bn, H, H, Cc, 0, ptr(w), None, CO, k, k, s, p, ptr(y), ptr(part), C.byref(mt), tile, ptr(TAIL), stream_ptr()))
    us = timeit(f); fl = 2*M*CO*k*k*Cc
    tiles = (-(-M//tile))*(-(-CO//tile))
    print(f"M={M:6d} N={CO:4d} K={k*k*Cc:5d} tiles={tiles:5d} ({tiles/256:5.2f}/CU): {us:7.1f} us {fl/us/1e6:6.1f} TF   us/tile-round {us/ -(-tiles//256):6.1f}", flush=True)
for Bn, H in ((16, 16), (32, 16), (48, 16), (49, 16), (64, 14), (64, 16), (80, 16), (96,16), (128, 16)):
    conv(Bn, H, 256, 256, 3, 1, 1)
