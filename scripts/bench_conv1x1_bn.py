"""Times conv1x1_fwd_bf3_bn (on-the-fly BatchNorm/residual/ReLU/split operand) against the plane route on ResNet shapes:
python3 scripts/bench_conv1x1_bn.py [--batch 64].  Per shape: planes conv alone, bn_apply-equivalent + conv is not timed here
(see scripts/agg_trace.py); fused with residual + fp32 copy / residual only / neither."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from depth_image_captioning_pub_amd import _lib  # noqa: E402
from depth_image_captioning_pub_amd._lib import check, ptr, stream_ptr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--fmt", type=int, default=0, help="operand format: 0 = bf16x3, 1 = f16x2")
ap.add_argument("--switches", default="", help="comma-separated dic_debug_force_staged_gemm codes")
ap.add_argument("--persist-grid", type=int, default=0, help="dic_conv_persistent_grid (0 = library default)")
ap.add_argument("--only", default="", help="run only the shapes whose name contains this")
ap.add_argument("--rotate", type=int, default=1, help="cycle through this many input / output buffer sets (> 256 MiB in total: no launch finds its operands in the Infinity Cache)")
a = ap.parse_args()
lib = _lib.load()
if a.persist_grid:
    check(lib.dic_conv_persistent_grid(a.persist_grid), "persistent grid")
for code in filter(None, a.switches.split(",")):
    assert lib.dic_debug_force_staged_gemm(int(code)) == 0, code
DEV = "cuda:0"


def split(x2d, scale=1.0):
    R, K = x2d.shape
    out = [torch.empty((R + 1) // 2 * 2 * K, dtype=torch.int16, device=DEV) for _ in range(3 - a.fmt)]
    if a.fmt:
        check(lib.dic_split_f16x2_paired(ptr(x2d), C.c_longlong(R), K, C.c_float(scale), ptr(out[0]), ptr(out[1]), stream_ptr()), "split")
        out.append(None)
    else:
        check(lib.dic_split_bf16x3_paired(ptr(x2d), C.c_longlong(R), K, ptr(out[0]), ptr(out[1]), ptr(out[2]), stream_ptr()), "split")
    return out


def planes(ps):
    return (C.c_void_p * 3)(*[t.data_ptr() if t is not None else None for t in ps])


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters * 1e3


B = a.batch
for name, M, Cin, CO in (("layer2 conv1", B * 784, 512, 128), ("layer2 conv3", B * 784, 128, 512), ("layer3 conv1", B * 196, 1024, 256),
                         ("layer3 conv3", B * 196, 256, 1024), ("layer4 conv1", B * 49, 2048, 512), ("layer4 conv3", B * 49, 512, 2048),
                         ("layer1 conv3", B * 3136, 64, 256)):
    if a.only not in name:
        continue
    raws = [torch.randn(M, Cin, device=DEV) for _ in range(a.rotate)]
    ress = [torch.randn(M, Cin, device=DEV) for _ in range(a.rotate)]
    outs = [torch.empty(M, Cin, device=DEV) for _ in range(a.rotate)]
    raw, res, out = raws[0], ress[0], outs[0]
    turn = [0]
    scale = torch.rand(Cin, device=DEV) + 0.5
    shift = torch.randn(Cin, device=DEV)
    w = torch.randn(CO, Cin, device=DEV) / Cin ** 0.5
    wp, ap_ = split(w, 16384.0), split(raw, 4.0)
    osc = C.c_float(1.0 / (4.0 * 16384.0))
    y = torch.empty(M, CO, device=DEV)
    part = torch.zeros((M // 64 + 2) * 2 * CO, device=DEV)
    tail = torch.empty(1024 * 64 * 64, device=DEV)
    mt = C.c_int(0)

    def plane_route():
        check(lib.dic_debug_conv_fmt(planes(ap_), 1, 1, M, Cin, planes(wp), CO, 1, 1, 0, ptr(y), ptr(part), C.byref(mt), ptr(tail), a.fmt, osc, stream_ptr()), "conv")

    def fused(r, o):
        def f():
            turn[0] = (turn[0] + 1) % a.rotate
            raw, res, out = raws[turn[0]], ress[turn[0]], outs[turn[0]]
            rc = lib.dic_debug_conv1x1_bn_fmt(ptr(raw), ptr(scale), ptr(shift), ptr(res) if r else None, 1, ptr(out) if o else None, M, Cin, planes(wp),
                                              CO, ptr(y), ptr(part), C.byref(mt), ptr(tail), 1024, a.fmt, osc, stream_ptr())
            assert rc in (0, 1), rc
            return rc
        return f

    t0 = timeit(plane_route)
    if a.fmt == 1 and Cin in (128, 256):        # conv3 shapes: the A-stationary kernel (raw input, BatchNorm-apply fused, no planes)
        part2 = torch.zeros((M // 32 + 4) * 2 * CO, device=DEV)

        def astat():
            rc = lib.dic_debug_conv1x1_astat(ptr(raw), ptr(scale), ptr(shift), 1, M, Cin, planes(wp), CO, ptr(y), ptr(part2), C.byref(mt), osc,
                                             None, stream_ptr())
            assert rc == 0, rc
        print(f"{name:13s} M={M} {Cin}->{CO}: A-stationary kernel {timeit(astat):7.1f} us")
    if fused(False, False)() == 1:
        print(f"{name:13s} M={M} {Cin}->{CO}: planes {t0:7.1f} us; not on the persistent kernel")
        continue
    t = [timeit(fused(r, o)) for r, o in ((True, True), (True, False), (False, False))]
    print(f"{name:13s} M={M} {Cin}->{CO}: planes {t0:7.1f} us; on-the-fly res+copy {t[0]:7.1f}, res {t[1]:7.1f}, plain {t[2]:7.1f} us")
