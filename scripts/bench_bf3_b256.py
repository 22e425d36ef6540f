"""bf16x3 kernel tile shapes on the batch-256 layer shapes of ResNet-152 (rowk microbenchmark, paired layout)."""
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
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, N, K) in [(50176, 256, 1024), (50176, 256, 2304), (50176, 1024, 256), (200704, 128, 1152), (200704, 512, 128), (200704, 128, 512), (12544, 512, 4608), (12544, 2048, 512), (12544, 512, 2048)]:
    A = torch.randn(M, K, device=DEV); B = torch.randn(N, K, device=DEV); a = split(A); b = split(B)
    Cm = torch.empty(M, N, device=DEV)
    f = lambda: check(lib.dic_gemm_bf16x3_paired(M, N, K, ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(b[0]), ptr(b[1]), ptr(b[2]), ptr(Cm), C.c_longlong(N), None, stream_ptr()))
    res = []
    for code in (11, 21, 22):
        lib.dic_debug_force_staged_gemm(code)
        for st in (42, 43):
            if code == 11 and st == 43: continue
            lib.dic_debug_force_staged_gemm(st)
            t = timeit(f); res.append(f"{code}/{st-40}st {t:7.1f}us {2*M*N*K/t/1e6:6.1f}TF")
    lib.dic_debug_force_staged_gemm(42); lib.dic_debug_force_staged_gemm(20)
    print(f"M={M:6d} N={N:5d} K={K:5d} | " + " | ".join(res), flush=True)
