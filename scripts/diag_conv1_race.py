"""Layer-1 kernels of the depth encoder alone, repeated on identical inputs while a second stream keeps the chip busy:
which outputs differ between repetitions?"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_image_captioning_pub_amd import _lib, synthetic as syn
from depth_image_captioning_pub_amd._lib import ptr
lib = _lib.load()
for code in sys.argv[1:]:
    lib.dic_debug_force_staged_gemm(int(code))
dev, B = "cuda:0", 64
x = syn.depth_maps(B, seed=123).to(dev)
g = torch.Generator(device="cpu").manual_seed(1)
dy = (torch.randn(B, 73, 73, 128, generator=g) * 1e-3).to(dev)
w = (torch.randn(128, 49, generator=g) * 0.1).to(dev); bias = torch.zeros(128, device=dev)
ws = torch.zeros(1024 * 6400, device=dev); cs = torch.zeros(256 * 2048 * 4, device=dev)
dw = torch.zeros(128 * 49, device=dev); db = torch.zeros(128, device=dev)
y = torch.zeros(B, 73, 73, 128, device=dev); part = torch.zeros(1100 * 2 * 128, device=dev)
hog_stream = torch.cuda.Stream()
a = torch.randn(4096, 4096, device=dev)
HOG = os.environ.get("HOG", "gemm")
if HOG.startswith("bf3gemm"):
    import ctypes as C
    from depth_image_captioning_pub_amd._lib import check
    def split(t):
        R, K = t.shape; Rp = (R + 1) // 2 * 2
        out = [torch.empty(Rp * K, dtype=torch.int16, device=dev) for _ in range(3)]
        check(lib.dic_split_bf16x3_paired(ptr(t), C.c_longlong(R), K, ptr(out[0]), ptr(out[1]), ptr(out[2]), C.c_void_p(torch.cuda.current_stream().cuda_stream))); return out
    M_, N_, K_ = 12544, 1024, 256
    A_ = torch.randn(M_, K_, device=dev); B_ = torch.randn(N_, K_, device=dev); a3 = split(A_); b3 = split(B_)
    Cm = torch.empty(M_, N_, device=dev)
    lib.dic_debug_force_staged_gemm(int(HOG[7:]) if len(HOG) > 7 else 11)
    torch.cuda.synchronize()
elif HOG != "gemm":
    from depth_image_captioning_pub_amd import native
    LAYERS = tuple(int(v) for v in os.environ.get("LAYERS", "3,8,36,3").split(","))
    rn = native.ResNetRunner({k: v.to(dev) for k, v in syn.resnet152_weights(seed=125, layers=LAYERS).items()}, LAYERS, conv_mode=HOG)
    imgs = syn.rgb_images(64, seed=123).to(dev)
    with torch.cuda.stream(hog_stream):
        rn.forward(imgs, True, compact=True)
    torch.cuda.synchronize()
main = torch.cuda.current_stream().cuda_stream
ref = {}
bad = {"dw": 0, "db": 0, "ws_rows": 0, "y": 0, "part": 0}
N = 60
for it in range(N):
    with torch.cuda.stream(hog_stream):
        if HOG == "gemm":
            for _ in range(6):
                a = (a @ a).clamp_(-1, 1)
        elif HOG.startswith("bf3gemm"):
            for _ in range(40):
                lib.dic_gemm_bf16x3_paired(M_, N_, K_, ptr(a3[0]), ptr(a3[1]), ptr(a3[2]), ptr(b3[0]), ptr(b3[1]), ptr(b3[2]), ptr(Cm), C.c_longlong(N_), None, C.c_void_p(hog_stream.cuda_stream))
        else:
            for _ in range(int(os.environ.get("REP", "1"))):
                rn.forward(imgs, True, compact=True)
    import time; time.sleep(0.004)            # let the side stream get going before the kernels under test are enqueued
    assert lib.dic_debug_conv1_fwd(ptr(x), B, 224, 224, ptr(w), ptr(bias), ptr(y), ptr(part), C.c_void_p(main)) == 0
    assert lib.dic_debug_conv1_wgrad(ptr(x), B, 224, 224, ptr(dy), ptr(dw), ptr(db), ptr(ws), ptr(cs), C.c_void_p(main)) == 0
    torch.cuda.synchronize()
    cur = {"dw": dw.clone(), "db": db.clone(), "ws": ws[:512 * 6400].clone().view(512, 6400), "y": y.clone(), "part": part[:512 * 256].clone()}
    if not ref:
        ref = cur
        print("run 0: nan in y", bool(torch.isnan(cur["y"]).any()), "nan in part", bool(torch.isnan(cur["part"]).any()), "nan in ws", bool(torch.isnan(cur["ws"]).any()))
        continue
    if it == 1:
        second = cur
    if it >= 2 and it < 6:
        print(f"it {it}: equal to run 1: y {torch.equal(cur['y'], second['y'])} dw {torch.equal(cur['dw'], second['dw'])}; y max|d| vs run0 {float((cur['y'] - ref['y']).abs().max()):.3e} "
              f"n differing {int((cur['y'] != ref['y']).sum())}; dw max|d| {float((cur['dw'] - ref['dw']).abs().max()):.3e} (scale {float(ref['dw'].abs().max()):.3e})")
    bad["dw"] += int(not torch.equal(cur["dw"], ref["dw"])); bad["db"] += int(not torch.equal(cur["db"], ref["db"]))
    bad["y"] += int(not torch.equal(cur["y"], ref["y"])); bad["part"] += int(not torch.equal(cur["part"], ref["part"]))
    rows = (cur["ws"] != ref["ws"]).any(dim=1).nonzero().flatten().tolist()
    if rows:
        bad["ws_rows"] += 1
        d = (cur["ws"] - ref["ws"])
        cols = (d[rows[0]] != 0).nonzero().flatten().tolist()
        print(f"it {it}: {len(rows)} partial rows differ (first {rows[:6]}); in row {rows[0]}: {len(cols)} columns, first {cols[:8]}, max |d| {float(d.abs().max()):.3e}", flush=True)
    if not torch.equal(cur["y"], ref["y"]):
        dd = (cur["y"] != ref["y"])
        idx = dd.nonzero()[:3].tolist()
        print(f"it {it}: y differs at {int(dd.sum())} elements, e.g. {idx}", flush=True)
print(HOG, bad)
if os.environ.get("PATTERN"):
    # one more repetition, described in detail
    with torch.cuda.stream(hog_stream):
        for _ in range(int(os.environ.get("REP", "1"))):
            rn.forward(imgs, True, compact=True)
    time.sleep(0.004)
    lib.dic_debug_conv1_fwd(ptr(x), B, 224, 224, ptr(w), ptr(bias), ptr(y), ptr(part), C.c_void_p(main)); torch.cuda.synchronize()
    dd = (y != ref["y"]).nonzero()
    print("differing elements", dd.shape[0])
    pix = torch.unique(dd[:, :3], dim=0)
    print("distinct pixels", pix.shape[0], "first", pix[:12].tolist())
    for pxl in pix[:6].tolist():
        b_, oh_, ow_ = pxl
        ch = (y[b_, oh_, ow_] != ref["y"][b_, oh_, ow_]).nonzero().flatten().tolist()
        print(" pixel", pxl, "row", b_ * 73 + oh_, "block", (b_ * 73 + oh_) % 512, "channels", len(ch), ch[:4], "..", ch[-2:],
              "max|d|", float((y[b_, oh_, ow_] - ref["y"][b_, oh_, ow_]).abs().max()))
    rows = torch.unique(dd[:, 0] * 73 + dd[:, 1])
    print("rows", rows.tolist()[:40])
