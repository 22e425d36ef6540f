"""Race screen for the persistent warp-specialised contraction kernel and the LDS-halo convolution: many repetitions of a few
shapes, alone and next to a bf16x3 ResNet forward on a second stream, every result compared bit for bit with the first one
(and the first one with the 64x64 kernel).  usage: stress_new_kernels.py [repetitions]"""
import sys, os, ctypes as C, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_image_captioning_pub_amd import _lib, native, synthetic as syn
from depth_image_captioning_pub_amd._lib import ptr, check
lib = _lib.load(); DEV = "cuda:0"
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 100
def sp(): return C.c_void_p(torch.cuda.current_stream().cuda_stream)
def split(x2d):
    R, K = x2d.shape
    out = [torch.empty((R + 1) // 2 * 2 * K, dtype=torch.int16, device=DEV) for _ in range(3)]
    check(lib.dic_split_bf16x3_paired(ptr(x2d), C.c_longlong(R), K, ptr(out[0]), ptr(out[1]), ptr(out[2]), sp()), "split")
    return out
hog_stream = torch.cuda.Stream()
LAYERS = (1, 1, 1, 1)
rn = native.ResNetRunner({k: v.to(DEV) for k, v in syn.resnet152_weights(seed=125, layers=LAYERS).items()}, LAYERS, conv_mode="bf16x3")
imgs = syn.rgb_images(64, seed=123).to(DEV)
with torch.cuda.stream(hog_stream):
    rn.forward(imgs, True, compact=True)
torch.cuda.synchronize()
def screen(name, launch, out, with_hog):
    ref, bad = None, 0
    for it in range(REPS):
        if with_hog:
            with torch.cuda.stream(hog_stream):
                for _ in range(3): rn.forward(imgs, True, compact=True)
            time.sleep(0.002)
        out.fill_(float("nan"))
        launch()
        torch.cuda.synchronize()
        if ref is None: ref = out.clone()
        elif not torch.equal(ref, out): bad += 1
    print(f"{name:46s} {'next to a forward' if with_hog else 'alone':18s}: {bad} of {REPS - 1} repetitions differ", flush=True)
    return ref
for (M, N, K) in [(50176, 256, 1024), (12544, 1024, 256), (33000, 256, 64), (4096, 1024, 1056)]:
    A = torch.randn(M, K, device=DEV); B = torch.randn(N, K, device=DEV); a = split(A); b = split(B)
    Cm = torch.empty(M, N, device=DEV)
    f = lambda: check(lib.dic_gemm_bf16x3_paired(M, N, K, ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(b[0]), ptr(b[1]), ptr(b[2]), ptr(Cm), C.c_longlong(N), None, sp()))
    lib.dic_debug_force_staged_gemm(11); f(); torch.cuda.synchronize(); base = Cm.clone()
    lib.dic_debug_force_staged_gemm(24)
    for hog in (False, True):
        r = screen(f"persistent ws {M}x{N}x{K}", f, Cm, hog)
        assert torch.equal(r, base), "differs from the 64x64 kernel"
    lib.dic_debug_force_staged_gemm(20)
planes = lambda ps: (C.c_void_p * 3)(*[t.data_ptr() for t in ps])
tail = torch.empty(256 * 64 * 64, device=DEV)
for (Bn, Cin, CO) in [(64, 256, 256), (256, 128, 128), (37, 64, 128)]:
    H = 14; M = Bn * H * H
    x = torch.randn(Bn, H, H, Cin, device=DEV); w = torch.randn(CO, 3, 3, Cin, device=DEV) / (9 * Cin) ** 0.5
    xp, wp = split(x.view(-1, Cin)), split(w.view(CO, -1))
    y = torch.empty(M, CO, device=DEV); part = torch.zeros((M // 64 + 2) * 2 * CO, device=DEV); mt = C.c_int(0)
    f = lambda: check(lib.dic_debug_conv_bf3(planes(xp), Bn, H, H, Cin, planes(wp), CO, 3, 1, 1, ptr(y), ptr(part), C.byref(mt), ptr(tail), sp()))
    lib.dic_debug_force_staged_gemm(74)
    for hog in (False, True):
        screen(f"halo conv B={Bn} C={Cin} CO={CO}", f, y, hog)
    lib.dic_debug_force_staged_gemm(78)
