import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_image_captioning_pub_amd import _lib
from depth_image_captioning_pub_amd._lib import ptr, stream_ptr, check
lib = _lib.load(); DEV = "cuda:0"
def split(x):
    n = x.numel(); hi = torch.empty(n, dtype=torch.int16, device=DEV); mid = torch.empty_like(hi); lo = torch.empty_like(hi)
    check(lib.dic_split_bf16x3(ptr(x), C.c_longlong(n), ptr(hi), ptr(mid), ptr(lo), stream_ptr())); return hi, mid, lo
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
def run(M, N, K, check_acc=True):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g); B = torch.randn(N, K, generator=g)
    Ad, Bd = A.to(DEV), B.to(DEV)
    ah, am, al = split(Ad); bh, bm, bl = split(Bd)
    # exactness of the split
    rec = (ah.view(torch.bfloat16).float() + am.view(torch.bfloat16).float()) + al.view(torch.bfloat16).float()
    exact = bool(torch.equal(rec.view(M, K), Ad))
    Cb = torch.empty(M, N, device=DEV); Cf = torch.empty(M, N, device=DEV)
    f3 = lambda: check(lib.dic_gemm_bf16x3(M, N, K, ptr(ah), ptr(am), ptr(al), C.c_longlong(K), ptr(bh), ptr(bm), ptr(bl), C.c_longlong(K), ptr(Cb), C.c_longlong(N), None, stream_ptr()))
    f1 = lambda: check(lib.dic_gemm_f32(M, N, K, ptr(Ad), C.c_longlong(K), 0, ptr(Bd), C.c_longlong(K), 0, ptr(Cf), C.c_longlong(N), None, 0, 0, 1, None, C.c_size_t(0), 64, stream_ptr()))
    t3, t1 = timeit(f3), timeit(f1)
    msg = f"M={M:6d} N={N:5d} K={K:5d}: bf16x3 {t3:8.1f} us {2*M*N*K/t3/1e6:6.1f} TF-eq | f32 MFMA {t1:8.1f} us {2*M*N*K/t1/1e6:6.1f} TF | split exact={exact}"
    if check_acc:
        ref = A.double() @ B.double().t()
        s = float(ref.abs().max())
        e3 = float((Cb.cpu().double() - ref).abs().max()) / s; e1 = float((Cf.cpu().double() - ref).abs().max()) / s
        r3 = float((Cb.cpu().double() - ref).norm() / ref.norm()); r1 = float((Cf.cpu().double() - ref).norm() / ref.norm())
        msg += f" | max-err/scale bf16x3 {e3:.2e} f32 {e1:.2e} | rel-l2 bf16x3 {r3:.2e} f32 {r1:.2e}"
    print(msg, flush=True)
run(256, 256, 256); run(1000, 130, 520); run(12544, 256, 2304); run(12544, 256, 1024); run(12544, 1024, 256)
run(4096, 4096, 4096, check_acc=False); run(50176, 256, 2304, check_acc=False)
