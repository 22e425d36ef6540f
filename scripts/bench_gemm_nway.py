"""The persistent warp-specialised bf16x3 GEMM on N streams at once, each launch limited to G workgroups: does the per-CU rate
hold when more CUs are busy?  Variants: full kernel / no DMA inside the loop (LDS reads + MFMA only: any slow-down is clock,
not memory) / cache-hot operands.  Run with GPU_MAX_HW_QUEUES=8 so that up to 8 streams get hardware queues of their own.
usage: python scripts/bench_gemm_nway.py [G] [M N K]"""
import os, sys, ctypes as C
os.environ.setdefault("DIC_LIB", "experiments")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_image_captioning_pub_amd import _lib
from depth_image_captioning_pub_amd._lib import ptr, check
lib = _lib.load(); DEV = "cuda:0"
G = int(sys.argv[1]) if len(sys.argv) > 1 else 49
M, N, K = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (12544, 1024, 256)
NMAX = 5


def split(x):
    R, K_ = x.shape
    out = [torch.empty((R + 1) // 2 * 2 * K_, dtype=torch.int16, device=DEV) for _ in range(3)]
    check(lib.dic_split_bf16x3_paired(ptr(x), C.c_longlong(R), K_, ptr(out[0]), ptr(out[1]), ptr(out[2]), _lib.stream_ptr()))
    return out


sets = []
for i in range(NMAX):
    A = torch.randn(M, K, device=DEV); B = torch.randn(N, K, device=DEV)
    sets.append((split(A), split(B), torch.empty(M, N, device=DEV)))
streams = [torch.cuda.Stream() for _ in range(NMAX)]
check(lib.dic_conv_persistent_grid(G))
lib.dic_debug_force_staged_gemm(24)
REPS = 40


def run(n):
    def go():
        for i in range(n):
            a, b, c = sets[i]
            sp = C.c_void_p(streams[i].cuda_stream)
            for _ in range(REPS):
                check(lib.dic_gemm_bf16x3_paired(M, N, K, ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(b[0]), ptr(b[1]), ptr(b[2]), ptr(c), C.c_longlong(N), None, sp))
    go(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda.synchronize()
    import time
    t0 = time.perf_counter(); go(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / REPS * 1e6


for abl, name in ((0, "full"), (1, "no DMA in loop"), (5, "A and B cache-hot")):
    lib.dic_debug_force_staged_gemm(50 + abl)
    base = run(1)
    for n in (1, 2, 3, 4, 5):
        t = run(n)
        print(f"G={G} {M}x{N}x{K} {name:18s}: {n} streams {t:8.1f} us per round (alone {base:.1f}; slow-down {t / base:.2f}x; {n * G} CUs asked)", flush=True)
lib.dic_debug_force_staged_gemm(50); lib.dic_debug_force_staged_gemm(20)
