import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_image_captioning_pub_amd import _lib
from depth_image_captioning_pub_amd._lib import ptr, stream_ptr, check
lib = _lib.load(); DEV = "cuda:0"
def split(x2d):
    R, K = x2d.shape
    out = [torch.empty((R + 1) // 2 * 2 * K, dtype=torch.int16, device=DEV) for _ in range(3)]
    check(lib.dic_split_bf16x3_paired(ptr(x2d), C.c_longlong(R), K, ptr(out[0]), ptr(out[1]), ptr(out[2]), stream_ptr()), "split")
    return out
planes = lambda ps: (C.c_void_p * 3)(*[p.data_ptr() for p in ps])
tail = torch.empty(256 * 64 * 64, device=DEV)
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (B, H, Cin, CO) in [(64, 24, 128, 512), (64, 7, 512, 2048)]:
    x = torch.randn(B, H, H, Cin, device=DEV); w = torch.randn(CO, 3, 3, Cin, device=DEV) / (9 * Cin) ** 0.5
    xp, wp = split(x.view(-1, Cin)), split(w.view(CO, -1))
    OH = H - 2; M = B * OH * OH
    y = torch.empty(M, CO, device=DEV); part = torch.zeros((M // 64 + 2) * 2 * CO, device=DEV); mt = C.c_int(0)
    f = lambda: check(lib.dic_debug_conv_bf3(planes(xp), B, H, H, Cin, planes(wp), CO, 3, 1, 0, ptr(y), ptr(part), C.byref(mt), ptr(tail), stream_ptr()))
    bias = torch.randn(CO, device=DEV)
    fb = lambda: check(lib.dic_conv2d_bf16x3(planes(xp), B, H, H, Cin, planes(wp), ptr(bias), CO, 3, 3, 1, 0, 0, ptr(y), ptr(tail), stream_ptr()))
    for code in (70, 79):
        lib.dic_debug_force_staged_gemm(code)
        print(f"   with bias (no statistics): policy {code}: {timeit(fb):.1f} us")
        print(f"B={B} {H}x{H}x{Cin} -> {CO}: policy {code}: {timeit(f):.1f} us (M={M}, tiles128 = {-(-M // 128) * (CO // 128)})")
