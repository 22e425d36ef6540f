"""3x3 convolution of 14x14 maps: the LDS-halo kernel (debug code 74) against the im2col gather kernels (75) - result vs an
fp64 convolution, BatchNorm partial sums, time per launch.  usage: bench_conv_halo.py [batch ...]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from depth_image_captioning_pub_amd import _lib
from depth_image_captioning_pub_amd._lib import ptr, stream_ptr, check
lib = _lib.load(); DEV = "cuda:0"
def split(x2d):
    R, K = x2d.shape
    out = [torch.empty((R + 1) // 2 * 2 * K, dtype=torch.int16, device=DEV) for _ in range(3)]
    check(lib.dic_split_bf16x3_paired(ptr(x2d), C.c_longlong(R), K, ptr(out[0]), ptr(out[1]), ptr(out[2]), stream_ptr()), "split")
    return out
def planes(ps):
    return (C.c_void_p * 3)(*[p.data_ptr() for p in ps])
tail = torch.empty(256 * 64 * 64, device=DEV)
def conv(xp, wp, B, H, Cin, CO, y, part):
    mt = C.c_int(0)
    check(lib.dic_debug_conv_bf3(planes(xp), B, H, H, Cin, planes(wp), CO, 3, 1, 1, ptr(y), ptr(part), C.byref(mt), ptr(tail), stream_ptr()), "conv")
    return mt.value
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for B in [int(a) for a in sys.argv[1:]] or [3, 64, 256]:
    for (Cin, CO) in [(256, 256), (64, 128)] if B > 3 else [(64, 128), (32, 256)]:
        H = 14
        g = torch.Generator().manual_seed(B + Cin)
        x = torch.randn(B, H, H, Cin, generator=g).to(DEV)
        w = (torch.randn(CO, 3, 3, Cin, generator=g) / (9 * Cin) ** 0.5).to(DEV)
        xp, wp = split(x.view(-1, Cin)), split(w.view(CO, -1))
        M = B * H * H
        res = {}
        for code in (75, 74):
            lib.dic_debug_force_staged_gemm(code)
            y = torch.full((M, CO), float("nan"), device=DEV); part = torch.zeros((M // 64 + 2) * 2 * CO, device=DEV)
            mt = conv(xp, wp, B, H, Cin, CO, y, part)
            t = timeit(lambda: conv(xp, wp, B, H, Cin, CO, y, part))
            res[code] = (y.clone(), part[: mt * 2 * CO].view(mt, 2, CO).double().sum(0), t)
        lib.dic_debug_force_staged_gemm(75)
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), padding=1).permute(0, 2, 3, 1).reshape(M, CO)
        sc = float(ref.abs().max())
        line = f"B={B:3d} C={Cin:3d} CO={CO:3d} M={M:6d}"
        for code in (75, 74):
            y, pt, t = res[code]
            e = float((y.double() - ref).abs().max()) / sc
            es = float((pt[0] - ref.sum(0)).abs().max() / ref.sum(0).abs().max()); es2 = float((pt[1] - (ref * ref).sum(0)).abs().max() / (ref * ref).sum(0).abs().max())
            line += f" | {'halo' if code == 74 else 'gather'}: {t:7.1f} us {2 * M * CO * 9 * Cin / t / 1e6:6.1f} TF-eq, err {e:.1e}, stats {es:.1e}/{es2:.1e}, finite {bool(torch.isfinite(y).all())}"
        print(line, flush=True)
        if os.environ.get("ABLATE"):
            lib.dic_debug_force_staged_gemm(74)
            y = torch.empty((M, CO), device=DEV); part = torch.zeros((M // 64 + 2) * 2 * CO, device=DEV)
            for abl, name in ((0, "full"), (1, "no weight DMA in loop"), (2, "no halo DMA in loop"), (3, "neither")):
                lib.dic_debug_force_staged_gemm(50 + abl)
                t = timeit(lambda: conv(xp, wp, B, H, Cin, CO, y, part))
                print(f"    halo kernel, {name:22s}: {t:7.1f} us", flush=True)
            lib.dic_debug_force_staged_gemm(50); lib.dic_debug_force_staged_gemm(75)
