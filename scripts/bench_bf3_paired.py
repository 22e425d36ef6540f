import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_image_captioning_pub_amd import _lib
from depth_image_captioning_pub_amd._lib import ptr, stream_ptr, check
lib = _lib.load(); DEV = "cuda:0"
def split(x):
    n = x.numel(); hi = torch.empty(n, dtype=torch.int16, device=DEV); mid = torch.empty_like(hi); lo = torch.empty_like(hi)
    check(lib.dic_split_bf16x3(ptr(x), C.c_longlong(n), ptr(hi), ptr(mid), ptr(lo), stream_ptr())); return hi, mid, lo
def pair(pl, R, K):   # [R,K] -> row-pair interleaved per 32-element block
    return pl.view(R // 2, 2, K // 32, 32).permute(0, 2, 1, 3).contiguous().view(-1)
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
lib.dic_debug_force_staged_gemm(11)
for (M, N, K) in [(4096, 4096, 4096), (12544, 256, 2304), (12544, 256, 1024), (12544, 1024, 256), (50176, 128, 512), (200704, 64, 256), (3136, 512, 2048)]:
    A = torch.randn(M, K, device=DEV); B = torch.randn(N, K, device=DEV); a = split(A); b = split(B)
    ap = [pair(x, M, K) for x in a]; bp = [pair(x, N, K) for x in b]
    C0 = torch.empty(M, N, device=DEV); C1 = torch.empty(M, N, device=DEV)
    
    f0 = lambda: check(lib.dic_gemm_bf16x3(M, N, K, ptr(a[0]), ptr(a[1]), ptr(a[2]), C.c_longlong(K), ptr(b[0]), ptr(b[1]), ptr(b[2]), C.c_longlong(K), ptr(C0), C.c_longlong(N), None, stream_ptr()))
    t0 = timeit(f0)
    
    f1 = lambda: check(lib.dic_gemm_bf16x3_paired(M, N, K, ptr(ap[0]), ptr(ap[1]), ptr(ap[2]), ptr(bp[0]), ptr(bp[1]), ptr(bp[2]), ptr(C1), C.c_longlong(N), None, stream_ptr()))
    t1 = timeit(f1)
    same = bool(torch.equal(C0, C1))
    print(f"M={M:6d} N={N:5d} K={K:5d} | plain {t0:7.1f}us {2*M*N*K/t0/1e6:6.1f}TF | paired {t1:7.1f}us {2*M*N*K/t1/1e6:6.1f}TF | identical={same}", flush=True)
