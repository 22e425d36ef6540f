"""GPU: the generic exact-fp32 MFMA contraction and the implicit-GEMM convolution, through the C ABI,
against torch CPU fp32 (the same ops the oracle is built from).  Tolerance: fp32 with a different
summation order -> relative 2e-5 of the output scale."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from depth_image_captioning_pub_amd._lib import check, ptr, stream_ptr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TAIL = None


def _close(got, ref, tol=2e-5):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    scale = float(ref.abs().max()) + 1e-12
    err = float((got - ref).abs().max())
    assert err <= tol * scale, f"max err {err:.3e} vs scale {scale:.3e}"


def _gemm(lib, A, B, a_colk, b_colk, M, N, K, bias=None, act=0, acc=None, splitk=1, tile=0):
    Cout = acc.clone() if acc is not None else torch.full((M, N), float("nan"), device=DEV)
    ws = torch.empty(max(1, splitk * M * N), device=DEV) if splitk > 1 else None
    rc = lib.dic_gemm_f32(M, N, K, ptr(A), C.c_longlong(A.stride(0)), a_colk, ptr(B), C.c_longlong(B.stride(0)),
                          b_colk, ptr(Cout), C.c_longlong(N), ptr(bias), act, 1 if acc is not None else 0, splitk,
                          ptr(ws), C.c_size_t(ws.numel() * 4 if ws is not None else 0), tile, stream_ptr())
    check(rc, "dic_gemm_f32")
    torch.cuda.synchronize()
    return Cout


@pytest.mark.parametrize("M,N,K,tile", [(128, 128, 64, 128), (300, 200, 100, 0), (64, 512, 2304, 64),
                                         (1000, 130, 50, 0), (257, 129, 33, 128), (5, 7, 3, 0), (1280, 1000, 128, 0)])
def test_gemm_rowk_rowk(lib, M, N, K, tile):
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g)
    bias = torch.randn(N, generator=g)
    ref = F.linear(A, B, bias)
    got = _gemm(lib, A.to(DEV), B.to(DEV), 0, 0, M, N, K, bias=bias.to(DEV), tile=tile)
    _close(got, ref)


def test_gemm_asymmetric_identity(lib):
    """A = I with an asymmetric B catches a transposed C write (guide section 3)."""
    n = 64
    A = torch.eye(n)
    B = torch.arange(n * n, dtype=torch.float32).reshape(n, n)      # B[n][k]
    got = _gemm(lib, A.to(DEV), B.to(DEV), 0, 0, n, n, n)
    assert torch.equal(got.cpu(), B.t().contiguous())


@pytest.mark.parametrize("M,N,K", [(130, 70, 200), (512, 2304, 1280), (50, 128, 27)])
def test_gemm_colk_variants(lib, M, N, K):
    g = torch.Generator().manual_seed(K)
    At = torch.randn(K, M, generator=g)      # A(i,k) = At[k,i]
    Bt = torch.randn(K, N, generator=g)
    A = torch.randn(M, K, generator=g)
    ref_tn = At.t() @ Bt
    _close(_gemm(lib, At.to(DEV), Bt.to(DEV), 1, 1, M, N, K), ref_tn)
    ref_nn = A @ Bt
    _close(_gemm(lib, A.to(DEV), Bt.to(DEV), 0, 1, M, N, K), ref_nn)
    Bn = torch.randn(N, K, generator=g)
    _close(_gemm(lib, At.to(DEV), Bn.to(DEV), 1, 0, M, N, K), At.t() @ Bn.t())


def test_gemm_splitk_accumulate_act(lib):
    g = torch.Generator().manual_seed(3)
    M, N, K = 64, 512, 2304
    A = torch.randn(M, K, generator=g) * 0.1
    B = torch.randn(N, K, generator=g) * 0.1
    bias = torch.randn(N, generator=g)
    ref = torch.sigmoid(F.linear(A, B, bias))
    _close(_gemm(lib, A.to(DEV), B.to(DEV), 0, 0, M, N, K, bias=bias.to(DEV), act=2, splitk=12), ref)
    C0 = torch.randn(M, N, generator=g)
    ref2 = C0 + A @ B.t()
    _close(_gemm(lib, A.to(DEV), B.to(DEV), 0, 0, M, N, K, acc=C0.to(DEV), splitk=5), ref2)
    _close(_gemm(lib, A.to(DEV), B.to(DEV), 0, 0, M, N, K, acc=C0.to(DEV)), ref2)


CONVS = [  # B,H,W,C,CO,k,s,p,nchw
    (2, 17, 19, 32, 64, 3, 1, 1, 0),
    (3, 14, 14, 64, 96, 1, 1, 0, 0),
    (2, 15, 15, 64, 128, 3, 2, 1, 0),
    (2, 16, 16, 128, 64, 1, 2, 0, 0),
    (2, 40, 40, 3, 64, 7, 2, 3, 1),      # ResNet stem (NCHW input, C=3)
    (2, 46, 46, 1, 128, 7, 3, 0, 1),     # depth-encoder conv1 (C=1, stride 3, no padding)
    (2, 12, 12, 128, 512, 3, 1, 0, 0),   # depth-encoder conv2 shape class
]


@pytest.mark.parametrize("cfg", CONVS)
@pytest.mark.parametrize("tile", [64, 128])
def test_conv_fwd_and_bn_partials(lib, cfg, tile):
    B, H, W, Cc, CO, k, s, p, nchw = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x = torch.randn(B, Cc, H, W, generator=g)
    w = torch.randn(CO, Cc, k, k, generator=g) / (Cc * k * k) ** 0.5
    bias = torch.randn(CO, generator=g)
    ref = F.conv2d(x, w, bias, stride=s, padding=p)                       # NCHW
    OH, OW = ref.shape[2], ref.shape[3]
    xin = x.to(DEV).contiguous() if nchw else x.permute(0, 2, 3, 1).contiguous().to(DEV)
    w_ohwi = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    y = torch.full((B, OH, OW, CO), float("nan"), device=DEV)
    M = B * OH * OW
    part = torch.zeros((M + 63) // 64 * 2 * CO, device=DEV)
    mt = C.c_int(0)
    global TAIL
    if TAIL is None:
        TAIL = torch.empty(256 * 64 * 64, device=DEV)
    rc = lib.dic_conv2d_fwd(ptr(xin), B, H, W, Cc, nchw, ptr(w_ohwi), ptr(bias.to(DEV)), CO, k, k, s, p, ptr(y),
                            ptr(part), C.byref(mt), tile, ptr(TAIL), stream_ptr())
    check(rc, "dic_conv2d_fwd")
    torch.cuda.synchronize()
    ref_nhwc = ref.permute(0, 2, 3, 1).contiguous()
    _close(y, ref_nhwc)
    pt = part[: mt.value * 2 * CO].reshape(mt.value, 2, CO).cpu().double().sum(0)
    flat = ref_nhwc.reshape(-1, CO).double()
    assert torch.allclose(pt[0], flat.sum(0), rtol=1e-4, atol=1e-3)
    assert torch.allclose(pt[1], (flat * flat).sum(0), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("B,H,C_in,CO,k", [(64, 14, 64, 256, 3), (13, 14, 64, 64, 1), (66, 14, 32, 128, 3)])
def test_conv_tail_split_matches_plain(lib, B, H, C_in, CO, k):
    """Shapes whose tile count leaves a small remainder modulo 256 take the remainder-tile K-split path
    (tail_fixup_kernel); results and BN partial sums must equal the plain launch up to summation order."""
    global TAIL
    if TAIL is None:
        TAIL = torch.empty(256 * 64 * 64, device=DEV)
    g = torch.Generator().manual_seed(B + CO)
    x = torch.randn(B, H, H, C_in, generator=g).to(DEV)
    w = (torch.randn(CO, k, k, C_in, generator=g) / (C_in * k * k) ** 0.5).to(DEV)
    bias = torch.randn(CO, generator=g).to(DEV)
    M = B * H * H
    outs = []
    for tail in (None, TAIL):
        y = torch.full((B, H, H, CO), float("nan"), device=DEV)
        part = torch.zeros((M + 63) // 64 * 2 * CO, device=DEV)
        mt = C.c_int(0)
        check(lib.dic_conv2d_fwd(ptr(x), B, H, H, C_in, 0, ptr(w), ptr(bias), CO, k, k, 1, k // 2, ptr(y), ptr(part),
                                 C.byref(mt), 64, ptr(tail), stream_ptr()), "dic_conv2d_fwd")
        torch.cuda.synchronize()
        outs.append((y, part[: mt.value * 2 * CO].reshape(mt.value, 2, CO).double().sum(0)))
    _close(outs[1][0], outs[0][0], 1e-5)
    assert torch.allclose(outs[1][1], outs[0][1], rtol=1e-5, atol=1e-3)
    ref = F.conv2d(x.permute(0, 3, 1, 2).cpu(), w.permute(0, 3, 1, 2).cpu(), bias.cpu(), padding=k // 2)
    _close(outs[1][0], ref.permute(0, 2, 3, 1), 2e-5)


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (1000, 130, 520), (777, 64, 2304), (64, 512, 32)])
def test_gemm_bf16x3_is_fp32_accurate(lib, M, N, K):
    """Split-bf16 contraction (csrc/gemm_bf3.hip): the hi/mid/lo split is exact and the result is at least as close to
    an fp64 evaluation as the exact-fp32 MFMA kernel (bound: 1.5x its error + 1 ulp-level slack)."""
    g = torch.Generator().manual_seed(M + K)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g) * torch.logspace(-3, 3, N).unsqueeze(1)      # wide dynamic range across rows
    ref = A.double() @ B.double().t()
    Ad, Bd = A.to(DEV), B.to(DEV)

    def split(x):
        hi, mid, lo = (torch.empty(x.numel(), dtype=torch.int16, device=DEV) for _ in range(3))
        check(lib.dic_split_bf16x3(ptr(x), C.c_longlong(x.numel()), ptr(hi), ptr(mid), ptr(lo), stream_ptr()), "split")
        rec = (hi.view(torch.bfloat16).float() + mid.view(torch.bfloat16).float()) + lo.view(torch.bfloat16).float()
        assert torch.equal(rec.view_as(x), x), "hi + mid + lo must reproduce the fp32 value exactly"
        return hi, mid, lo

    a, b = split(Ad), split(Bd)
    C3 = torch.full((M, N), float("nan"), device=DEV)
    check(lib.dic_gemm_bf16x3(M, N, K, ptr(a[0]), ptr(a[1]), ptr(a[2]), C.c_longlong(K), ptr(b[0]), ptr(b[1]), ptr(b[2]),
                              C.c_longlong(K), ptr(C3), C.c_longlong(N), None, stream_ptr()), "dic_gemm_bf16x3")
    C1 = _gemm(lib, Ad, Bd, 0, 0, M, N, K)
    col = ref.abs().max(dim=0).values + 1e-30                                        # per-column scale
    e3 = float(((C3.cpu().double() - ref).abs() / col).max())
    e1 = float(((C1.cpu().double() - ref).abs() / col).max())
    assert e3 <= 1.5 * e1 + 2e-7, (e3, e1)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(128, 64, 64), (197, 131, 96), (1001, 65, 288), (3136, 256, 576)])
def test_gemm_bf16x3_paired_layout_is_bit_identical(lib, M, N, K):
    """The row-pair interleaved plane layout (include/dic.h, dic_split_bf16x3_paired: the format the convolutions
    consume) is a pure relayout: planes hold the same bf16 values at the documented offsets (a zero row pads an odd row
    count) and the contraction result is bit-identical to the plain-layout kernel, ragged M / N included."""
    g = torch.Generator().manual_seed(7 * M + K)
    A = torch.randn(M, K, generator=g).to(DEV)
    B = torch.randn(N, K, generator=g).to(DEV)

    def split_plain(x):
        out = [torch.empty(x.numel(), dtype=torch.int16, device=DEV) for _ in range(3)]
        check(lib.dic_split_bf16x3(ptr(x), C.c_longlong(x.numel()), ptr(out[0]), ptr(out[1]), ptr(out[2]), stream_ptr()), "split")
        return out

    def split_paired(x):
        R = x.shape[0]
        Rp = (R + 1) // 2 * 2
        out = [torch.full((Rp * K,), 0x7fff, dtype=torch.int16, device=DEV) for _ in range(3)]
        check(lib.dic_split_bf16x3_paired(ptr(x), C.c_longlong(R), K, ptr(out[0]), ptr(out[1]), ptr(out[2]), stream_ptr()),
              "split_paired")
        return out

    ap, bp = split_plain(A), split_plain(B)
    aq, bq = split_paired(A), split_paired(B)
    for plain, paired, R in ((ap, aq, M), (bp, bq, N)):
        Rp = (R + 1) // 2 * 2
        for pl, pq in zip(plain, paired):
            want = torch.zeros(Rp, K, dtype=torch.int16, device=DEV)
            want[:R] = pl.view(R, K)
            want = want.view(Rp // 2, 2, K // 32, 32).permute(0, 2, 1, 3).reshape(-1)
            assert torch.equal(pq, want)
    C0 = torch.full((M, N), float("nan"), device=DEV)
    C1 = torch.full((M, N), float("nan"), device=DEV)
    check(lib.dic_gemm_bf16x3(M, N, K, ptr(ap[0]), ptr(ap[1]), ptr(ap[2]), C.c_longlong(K), ptr(bp[0]), ptr(bp[1]), ptr(bp[2]),
                              C.c_longlong(K), ptr(C0), C.c_longlong(N), None, stream_ptr()), "dic_gemm_bf16x3")
    check(lib.dic_gemm_bf16x3_paired(M, N, K, ptr(aq[0]), ptr(aq[1]), ptr(aq[2]), ptr(bq[0]), ptr(bq[1]), ptr(bq[2]),
                                     ptr(C1), C.c_longlong(N), None, stream_ptr()), "dic_gemm_bf16x3_paired")
    assert torch.isfinite(C1).all()
    assert torch.equal(C0, C1)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (1000, 300, 96), (129, 130, 64), (40000, 384, 160), (33000, 256, 64),
                                   (4096, 1024, 1056)])
def test_gemm_bf16x3_tile_variants_are_bit_identical(lib, M, N, K):
    """Every workgroup-tile variant of the split-bf16 kernel that the product library contains sums the six products of a K
    index in the same order, K ascending: 64x64 (code 11), 128x64 (21) and the persistent warp-specialised 128x128 kernel
    (24: one workgroup per CU streams the K tiles of all its output tiles through a 3-stage ring filled by four producer
    waves, stores at the seams) must agree bit for bit - ragged edges, a single tile, many seams per workgroup
    (33000 x 256 x 64 = 516 tiles of two K tiles on 224 workgroups) and long K included.  (The parked forms - deep-pipelined,
    computing-wave DMA, 256x128 - are not in libdic_hip.so; scripts/experiments/test_parked_kernels_gpu.py covers them.)"""
    g = torch.Generator().manual_seed(3 * M + K)
    A = torch.randn(M, K, generator=g).to(DEV)
    B = (torch.randn(N, K, generator=g) * torch.logspace(-2, 2, N).unsqueeze(1)).to(DEV)

    def split_paired(x):
        R = x.shape[0]
        out = [torch.empty((R + 1) // 2 * 2 * K, dtype=torch.int16, device=DEV) for _ in range(3)]
        check(lib.dic_split_bf16x3_paired(ptr(x), C.c_longlong(R), K, ptr(out[0]), ptr(out[1]), ptr(out[2]), stream_ptr()),
              "split_paired")
        return out

    a, b = split_paired(A), split_paired(B)
    outs = {}
    try:
        for code in (11, 21, 24):
            assert lib.dic_debug_force_staged_gemm(code) == 0
            for rep in range(3):               # the persistent kernel's hand-offs are timing dependent: repeat
                Cm = torch.full((M, N), float("nan"), device=DEV)
                check(lib.dic_gemm_bf16x3_paired(M, N, K, ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(b[0]), ptr(b[1]), ptr(b[2]),
                                                 ptr(Cm), C.c_longlong(N), None, stream_ptr()), "dic_gemm_bf16x3_paired")
                assert torch.isfinite(Cm).all(), code
                if code in outs:
                    assert torch.equal(outs[code], Cm), f"code {code}: repetition {rep} differs"
                outs[code] = Cm
    finally:
        lib.dic_debug_force_staged_gemm(20)
    for code in (21, 24):
        assert torch.equal(outs[11], outs[code]), f"tile variant {code} differs from the 64x64 kernel"
    ref = A.double() @ B.double().t()
    col = ref.abs().max(dim=0).values + 1e-30
    assert float(((outs[24].double() - ref).abs() / col).max()) < 5e-6


@pytest.mark.gpu
@pytest.mark.parametrize("B,Cin,CO", [(3, 64, 128), (5, 32, 256), (40, 96, 128), (64, 256, 256)])
def test_conv3x3_halo_kernel_vs_fp64_and_gather(lib, B, Cin, CO):
    """3x3 / stride 1 / pad 1 convolution of 14x14 maps (ResNet layer 3) on the LDS-halo kernel (csrc/gemm_bf3.hip,
    conv3x3_bf3_halo_kernel): output and train-mode BatchNorm partial sums against an fp64 convolution, with the im2col gather
    kernel's own error as the yardstick (chunk-major instead of tap-major summation: same products, not bit-identical).
    Covers a ragged last tile (588 and 980 output pixels), tiles that span two and three images, one and eight channel chunks,
    several tiles per workgroup, and repeats each launch (the kernel's DMA hand-offs are timing dependent)."""
    import torch.nn.functional as F
    H = 14
    g = torch.Generator().manual_seed(B + Cin)
    x = torch.randn(B, H, H, Cin, generator=g).to(DEV)
    w = (torch.randn(CO, 3, 3, Cin, generator=g) / (9 * Cin) ** 0.5).to(DEV)

    def split(x2d):
        R, K = x2d.shape
        out = [torch.empty((R + 1) // 2 * 2 * K, dtype=torch.int16, device=DEV) for _ in range(3)]
        check(lib.dic_split_bf16x3_paired(ptr(x2d), C.c_longlong(R), K, ptr(out[0]), ptr(out[1]), ptr(out[2]), stream_ptr()), "split")
        return out

    xp, wp = split(x.view(-1, Cin)), split(w.view(CO, -1))
    planes = lambda ps: (C.c_void_p * 3)(*[t.data_ptr() for t in ps])                    # noqa: E731
    M = B * H * H
    global TAIL
    if TAIL is None:
        TAIL = torch.empty(256 * 64 * 64, device=DEV)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.permute(0, 3, 1, 2).double().cpu(), padding=1).permute(0, 2, 3, 1).reshape(M, CO)
    scale = float(ref.abs().max())
    res = {}
    try:
        for code in (75, 74):                                   # gather kernels, halo kernel
            lib.dic_debug_force_staged_gemm(code)
            for rep in range(3):
                y = torch.full((M, CO), float("nan"), device=DEV)
                part = torch.zeros((M // 64 + 2) * 2 * CO, device=DEV)
                mt = C.c_int(0)
                check(lib.dic_debug_conv_bf3(planes(xp), B, H, H, Cin, planes(wp), CO, 3, 1, 1, ptr(y), ptr(part), C.byref(mt), ptr(TAIL),
                                             stream_ptr()), "dic_debug_conv_bf3")
                torch.cuda.synchronize()
                assert torch.isfinite(y).all()
                if code in res:
                    assert torch.equal(res[code][0], y), f"code {code}: repetition {rep} differs"
                res[code] = (y, part[: mt.value * 2 * CO].view(mt.value, 2, CO).double().sum(0).cpu())
    finally:
        lib.dic_debug_force_staged_gemm(78)
    err = {c: float((res[c][0].double().cpu() - ref).abs().max()) / scale for c in res}
    assert err[74] <= 2.0 * err[75] + 1e-6, err
    for c in res:
        assert torch.allclose(res[c][1][0], ref.sum(0), rtol=1e-4, atol=1e-3 * scale), c
        assert torch.allclose(res[c][1][1], (ref * ref).sum(0), rtol=1e-4, atol=1e-3 * scale), c


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 3])
def test_persistent_kernels_remainder_round_k_split(lib, k):
    """Remainder-round K split of the persistent 128x128 kernels (csrc/gemm_bf3.hip, launch_bf3).  3x3 (LDS-halo kernel): 274
    output tiles (89 images of 14x14, 256 -> 256 channels; the last tile ragged) = one full round of 256 workgroups + 18
    remainder tiles.  1x1 (warp-specialised kernel; ResNet layer 3 conv3 at batch 64): 784 tiles (64 images, 256 -> 1024) =
    three rounds + 16.  The remainder tiles are cut into K slices (1x1: 4 slices of two K tiles; 3x3: 3 slices of channel
    chunks - the 4-MB tail workspace holds 256 [64][64] slabs, four per piece) and finished by the tail fix-up in its 128x128 mode.  Against the same launch without the split (code 90) and an fp64 convolution:
    output error <= 2x the unsplit kernel's, train-mode BatchNorm partial sums = column sums / sums of squares of the stored
    output, bit-reproducible over repetitions, and the slices really went through the workspace."""
    import torch.nn.functional as F
    B, H, Cin, CO = (89, 14, 256, 256) if k == 3 else (64, 14, 256, 1024)
    rem, slices = (18, 3) if k == 3 else (16, 4)              # remainder tiles of the 256-workgroup grid, K slices each
    g = torch.Generator().manual_seed(70 + k)
    x = torch.randn(B, H, H, Cin, generator=g).to(DEV)
    w = (torch.randn(CO, k, k, Cin, generator=g) / (k * k * Cin) ** 0.5).to(DEV)

    def split(x2d):
        R, K = x2d.shape
        out = [torch.empty((R + 1) // 2 * 2 * K, dtype=torch.int16, device=DEV) for _ in range(3)]
        check(lib.dic_split_bf16x3_paired(ptr(x2d), C.c_longlong(R), K, ptr(out[0]), ptr(out[1]), ptr(out[2]), stream_ptr()), "split")
        return out

    xp, wp = split(x.view(-1, Cin)), split(w.view(CO, -1))
    planes = lambda ps: (C.c_void_p * 3)(*[t.data_ptr() for t in ps])                    # noqa: E731
    M = B * H * H
    ref = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.permute(0, 3, 1, 2).double().cpu(), padding=k // 2).permute(0, 2, 3, 1).reshape(M, CO)
    scale = float(ref.abs().max())
    tail = torch.full((256 * 64 * 64 + 4096,), float("nan"), device=DEV)      # the documented 4 MB + a guard slab
    res, touched = {}, {}
    try:
        for code in (90, 91):                                   # remainder split off / on
            assert lib.dic_debug_force_staged_gemm(code) == 0
            tail.fill_(float("nan"))
            for rep in range(3):
                y = torch.full((M, CO), float("nan"), device=DEV)
                part = torch.zeros((M // 64 + 2) * 2 * CO, device=DEV)
                mt = C.c_int(0)
                check(lib.dic_debug_conv_bf3(planes(xp), B, H, H, Cin, planes(wp), CO, k, 1, k // 2, ptr(y), ptr(part), C.byref(mt), ptr(tail),
                                             stream_ptr()), "dic_debug_conv_bf3")
                torch.cuda.synchronize()
                assert torch.isfinite(y).all()
                stats = part[: mt.value * 2 * CO].view(mt.value, 2, CO).double().sum(0).cpu()
                if code in res:
                    assert torch.equal(res[code][0], y), f"code {code}: repetition {rep} differs"
                res[code] = (y, stats)
            touched[code] = int(torch.isfinite(tail).sum())
    finally:
        lib.dic_debug_force_staged_gemm(91)
    assert touched[90] == 0 and touched[91] == rem * 4 * slices * 64 * 64, touched      # remainder tiles x 4 quadrants x slices
    assert not torch.isfinite(tail[256 * 64 * 64:]).any(), "wrote past the 4-MB tail workspace"
    err = {c: float((res[c][0].double().cpu() - ref).abs().max()) / scale for c in res}
    assert err[91] <= 2.0 * err[90] + 1e-6 and err[91] < 5e-6, err
    ntn = CO // 128
    first_rem_row = ((M + 127) // 128 * ntn - rem) // ntn * 128      # rows of the tiles finished whole (tile = row block x column block)
    assert torch.equal(res[90][0][:first_rem_row], res[91][0][:first_rem_row])      # whole tiles: the same kernel code, bit for bit
    for c in res:
        yd = res[c][0].double().cpu()
        assert torch.allclose(res[c][1][0], yd.sum(0), rtol=1e-5, atol=1e-4 * scale), c
        assert torch.allclose(res[c][1][1], (yd * yd).sum(0), rtol=1e-5, atol=1e-4 * scale), c


@pytest.mark.gpu
@pytest.mark.parametrize("M,Cin,CO,res,relu", [(12544, 1024, 256, True, 1), (12544, 256, 1024, False, 1), (12500, 512, 256, True, 0),
                                               (50176, 512, 128, False, 1), (200704, 64, 256, True, 1)])
def test_conv1x1_with_on_the_fly_bn_operand(lib, M, Cin, CO, res, relu):
    """conv1x1_fwd_bf3_bn (csrc/gemm_bf3.hip): the persistent warp-specialised kernel whose producer waves form the A operand
    on the fly - act(raw * scale + shift (+ residual)), ReLU, exact three-plane bf16 split - instead of reading planes that a
    bn_apply_planes pass wrote.  Against the two-step route (torch for the elementwise part, dic_split_bf16x3_paired,
    dic_debug_conv_bf3) and an fp64 evaluation: ResNet layer-3 shapes (conv1 with residual, conv3 = 784 tiles with remainder
    pieces), a ragged last tile, a short-K layer-1 shape; the materialised fp32 activation (act_out), the BatchNorm partial sums,
    bit-reproducibility over repetitions, and the 'not eligible' answer for a shape the policy keeps off the persistent kernel."""
    g = torch.Generator().manual_seed(M + Cin)
    raw = torch.randn(M, Cin, generator=g).to(DEV)
    scale = (torch.rand(Cin, generator=g) + 0.5).to(DEV)
    shift = (torch.randn(Cin, generator=g) * 0.3).to(DEV)
    resid = torch.randn(M, Cin, generator=g).to(DEV) if res else None
    w = (torch.randn(CO, Cin, generator=g) / Cin ** 0.5).to(DEV)
    act = torch.addcmul(shift, raw, scale)                      # raw * scale + shift (fused multiply-add, as the kernels contract it)
    if res:
        act = act + resid
    if relu:
        act = torch.relu(act)

    def split(x2d):
        R, K = x2d.shape
        out = [torch.empty((R + 1) // 2 * 2 * K, dtype=torch.int16, device=DEV) for _ in range(3)]
        check(lib.dic_split_bf16x3_paired(ptr(x2d), C.c_longlong(R), K, ptr(out[0]), ptr(out[1]), ptr(out[2]), stream_ptr()), "split")
        return out

    planes = lambda ps: (C.c_void_p * 3)(*[t.data_ptr() for t in ps])                    # noqa: E731
    wp = split(w)
    tail = torch.empty(1024 * 64 * 64, device=DEV)
    # ---- two-step route through the plane format
    ap = split(act.contiguous())
    y_ref = torch.full((M, CO), float("nan"), device=DEV)
    part_ref = torch.zeros((M // 64 + 2) * 2 * CO, device=DEV)
    mt = C.c_int(0)
    check(lib.dic_debug_conv_bf3(planes(ap), 1, 1, M, Cin, planes(wp), CO, 1, 1, 0, ptr(y_ref), ptr(part_ref), C.byref(mt), ptr(tail),
                                 stream_ptr()), "dic_debug_conv_bf3")
    # ---- on-the-fly operand
    outs = []
    for rep in range(3):
        y = torch.full((M, CO), float("nan"), device=DEV)
        a_out = torch.full((M, Cin), float("nan"), device=DEV)
        part = torch.zeros((M // 64 + 2) * 2 * CO, device=DEV)
        mt2 = C.c_int(0)
        rc = lib.dic_debug_conv1x1_bn(ptr(raw), ptr(scale), ptr(shift), ptr(resid), relu, ptr(a_out), M, Cin, planes(wp), CO, ptr(y),
                                      ptr(part), C.byref(mt2), ptr(tail), 1024, stream_ptr())
        assert rc == 0, (rc, lib.dic_last_error())
        torch.cuda.synchronize()
        assert torch.isfinite(y).all() and torch.isfinite(a_out).all()
        outs.append((y, a_out, part[: mt2.value * 2 * CO].view(mt2.value, 2, CO).double().sum(0).cpu()))
        if rep:
            assert torch.equal(outs[0][0], y) and torch.equal(outs[0][1], a_out), f"repetition {rep} differs"
    y, a_out, stats = outs[0]
    assert float((a_out - act).abs().max()) <= 1e-6 * float(act.abs().max()), "materialised activation"
    ref64 = act.double().cpu() @ w.double().cpu().t()
    sc = float(ref64.abs().max())
    e_fused, e_two = float((y.double().cpu() - ref64).abs().max()) / sc, float((y_ref.double().cpu() - ref64).abs().max()) / sc
    print(f"\n{M}x{CO}x{Cin}: max err / scale vs fp64: on-the-fly operand {e_fused:.2e}, plane route {e_two:.2e}; "
          f"bit-identical outputs: {bool(torch.equal(y, y_ref))}")
    assert e_fused <= 2.0 * e_two + 1e-6 and e_fused < 5e-6
    yd = y.double().cpu()
    assert torch.allclose(stats[0], yd.sum(0), rtol=1e-5, atol=1e-4 * sc) and torch.allclose(stats[1], (yd * yd).sum(0), rtol=1e-5, atol=1e-4 * sc)
    # a shape the launch policy does not put on the persistent kernel (64 output channels): nothing is launched, the caller is told
    w64 = split(w[:64].contiguous())
    y64 = torch.full((M, 64), float("nan"), device=DEV)
    rc = lib.dic_debug_conv1x1_bn(ptr(raw), ptr(scale), ptr(shift), None, 1, None, M, Cin, planes(w64), 64, ptr(y64), None, None, ptr(tail),
                                  1024, stream_ptr())
    torch.cuda.synchronize()
    assert rc == 1 and torch.isnan(y64).all()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["1x1 persistent", "1x1 persistent remainder", "3x3 halo", "3x3 gathered persistent", "1x1 64-wide", "3x3 64-wide strided",
                                  "1x1 ragged", "1x1 few tiles long K", "3x3 few tiles long K", "1x1 256x128 tiles ragged", "1x1 256x128 tiles short K"])
def test_conv_f16x2_operand_format(lib, case):
    """The two-term fp16 operand format (dic_split_f16x2_paired; three products h1*h1' + h1*h2' + h2*h1', epilogue unscale) on every
    convolution kernel family of the ResNet forward, against an fp64 evaluation and beside the bf16x3 route on the same inputs: the
    error may be a few times the bf16x3 one (2^-22 representation error per operand instead of an exact split) but must stay at
    fp32-rounding level (< 4e-6 of the output scale; K up to 2304 here); BatchNorm partial sums agree with the stored output."""
    B, H, W, Cin, CO, k, stride, pad = {"1x1 persistent": (64, 14, 14, 1024, 256, 1, 1, 0), "1x1 persistent remainder": (64, 14, 14, 256, 1024, 1, 1, 0),
                                        "3x3 halo": (64, 14, 14, 256, 256, 3, 1, 1), "3x3 gathered persistent": (64, 28, 28, 128, 128, 3, 1, 1),
                                        "1x1 64-wide": (8, 14, 14, 512, 64, 1, 1, 0), "3x3 64-wide strided": (16, 28, 28, 128, 128, 3, 2, 1),
                                        "1x1 ragged": (63, 14, 14, 512, 256, 1, 1, 0),
                                        # ResNet layer 4 at batch 64: 100 / 100 output tiles of 128x128, every one cut into K slices
                                        "1x1 few tiles long K": (64, 7, 7, 2048, 512, 1, 1, 0), "3x3 few tiles long K": (64, 7, 7, 512, 512, 3, 1, 1),
                                        # f16x2: >= 192 tiles of 256x128 -> the twelve-wave kernel (a ragged last tile; K = 64: two K tiles per tile)
                                        "1x1 256x128 tiles ragged": (63, 14, 14, 256, 1024, 1, 1, 0), "1x1 256x128 tiles short K": (16, 56, 56, 64, 256, 1, 1, 0)}[case]
    g = torch.Generator().manual_seed(len(case))
    x = torch.relu(torch.randn(B, H, W, Cin, generator=g) * 1.5 + 0.3).to(DEV)         # post-ReLU-like activations
    w = (torch.randn(CO, k, k, Cin, generator=g) / (k * k * Cin) ** 0.5).to(DEV)
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    M = B * OH * OW
    ref64 = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), w.double().cpu().permute(0, 3, 1, 2), stride=stride, padding=pad).permute(0, 2, 3, 1).reshape(M, CO)
    sc = float(ref64.abs().max())

    def planes(ps):
        return (C.c_void_p * 3)(*[t.data_ptr() if t is not None else None for t in ps])

    def split(x2d, fmt, scale=1.0):
        R, K = x2d.shape
        out = [torch.zeros((R + 1) // 2 * 2 * K, dtype=torch.int16, device=DEV) for _ in range(3 - fmt)]
        if fmt:
            check(lib.dic_split_f16x2_paired(ptr(x2d), C.c_longlong(R), K, C.c_float(scale), ptr(out[0]), ptr(out[1]), stream_ptr()), "split f16x2")
            out.append(None)
        else:
            check(lib.dic_split_bf16x3_paired(ptr(x2d), C.c_longlong(R), K, ptr(out[0]), ptr(out[1]), ptr(out[2]), stream_ptr()), "split")
        return out

    tail = torch.empty(1024 * 64 * 64, device=DEV)
    w_scale = 2.0 ** math.floor(14 - math.log2(float(w.abs().max())))
    errs = {}
    for fmt in (0, 1):
        xp = split(x.reshape(B * H * W, Cin), fmt, 4.0)
        wp = split(w.reshape(CO, k * k * Cin), fmt, w_scale)
        y = torch.full((M, CO), float("nan"), device=DEV)
        part = torch.zeros((M // 64 + 2) * 2 * CO, device=DEV)
        mt = C.c_int(0)
        check(lib.dic_debug_conv_fmt(planes(xp), B, H, W, Cin, planes(wp), CO, k, stride, pad, ptr(y), ptr(part), C.byref(mt), ptr(tail), fmt,
                                     C.c_float(1.0 / (4.0 * w_scale)), stream_ptr()), "dic_debug_conv_fmt")
        torch.cuda.synchronize()
        assert torch.isfinite(y).all()
        yd = y.double().cpu()
        errs[fmt] = float((yd - ref64).abs().max()) / sc
        stats = part[: mt.value * 2 * CO].view(mt.value, 2, CO).double().sum(0).cpu()
        assert torch.allclose(stats[0], yd.sum(0), rtol=1e-5, atol=1e-4 * sc) and torch.allclose(stats[1], (yd * yd).sum(0), rtol=1e-5, atol=1e-4 * sc)
    print(f"\n{case}: max err / scale vs fp64: bf16x3 {errs[0]:.2e}, f16x2 {errs[1]:.2e}")
    assert errs[0] < 2e-6 and errs[1] < 4e-6


@pytest.mark.gpu
@pytest.mark.parametrize("M,Cin,CO,res", [(12544, 1024, 256, True), (12544, 256, 1024, False), (12500, 512, 256, True), (3136, 2048, 512, True)])
def test_conv1x1_on_the_fly_operand_f16x2(lib, M, Cin, CO, res):
    """conv1x1_fwd_bf3_bn in the f16x2 format: the producer waves scale the activation by 4 and write two fp16 planes; against fp64 and
    against the plane route of the same format (dic_split_f16x2_paired of the torch-evaluated activation)."""
    g = torch.Generator().manual_seed(M + Cin + 1)
    raw = torch.randn(M, Cin, generator=g).to(DEV)
    scale = (torch.rand(Cin, generator=g) + 0.5).to(DEV)
    shift = (torch.randn(Cin, generator=g) * 0.3).to(DEV)
    resid = torch.randn(M, Cin, generator=g).to(DEV) if res else None
    w = (torch.randn(CO, Cin, generator=g) / Cin ** 0.5).to(DEV)
    act = torch.addcmul(shift, raw, scale)
    if res:
        act = act + resid
    act = torch.relu(act)
    w_scale = 2.0 ** math.floor(14 - math.log2(float(w.abs().max())))
    wp = [torch.zeros((CO + 1) // 2 * 2 * Cin, dtype=torch.int16, device=DEV) for _ in range(2)]
    check(lib.dic_split_f16x2_paired(ptr(w), C.c_longlong(CO), Cin, C.c_float(w_scale), ptr(wp[0]), ptr(wp[1]), stream_ptr()), "split w")
    wpl = (C.c_void_p * 3)(wp[0].data_ptr(), wp[1].data_ptr(), None)
    tail = torch.empty(1024 * 64 * 64, device=DEV)
    y = torch.full((M, CO), float("nan"), device=DEV)
    a_out = torch.full((M, Cin), float("nan"), device=DEV)
    part = torch.zeros((M // 64 + 2) * 2 * CO, device=DEV)
    mt = C.c_int(0)
    rc = lib.dic_debug_conv1x1_bn_fmt(ptr(raw), ptr(scale), ptr(shift), ptr(resid), 1, ptr(a_out), M, Cin, wpl, CO, ptr(y), ptr(part), C.byref(mt),
                                      ptr(tail), 1024, 1, C.c_float(1.0 / (4.0 * w_scale)), stream_ptr())
    assert rc == 0, (rc, lib.dic_last_error())
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    assert float((a_out - act).abs().max()) <= 1e-6 * float(act.abs().max())
    ref64 = act.double().cpu() @ w.double().cpu().t()
    sc = float(ref64.abs().max())
    err = float((y.double().cpu() - ref64).abs().max()) / sc
    print(f"\n{M}x{CO}x{Cin} f16x2 on-the-fly operand: max err / scale vs fp64 {err:.2e}")
    assert err < 4e-6
    yd = y.double().cpu()
    stats = part[: mt.value * 2 * CO].view(mt.value, 2, CO).double().sum(0).cpu()
    assert torch.allclose(stats[0], yd.sum(0), rtol=1e-5, atol=1e-4 * sc) and torch.allclose(stats[1], (yd * yd).sum(0), rtol=1e-5, atol=1e-4 * sc)
    # eight producer waves (default, switch 113) and four (112) are the same arithmetic on another thread mapping: bit-identical
    try:
        assert lib.dic_debug_force_staged_gemm(112) == 0
        y4 = torch.full((M, CO), float("nan"), device=DEV)
        a4 = torch.full((M, Cin), float("nan"), device=DEV)
        rc = lib.dic_debug_conv1x1_bn_fmt(ptr(raw), ptr(scale), ptr(shift), ptr(resid), 1, ptr(a4), M, Cin, wpl, CO, ptr(y4), ptr(part), C.byref(mt),
                                          ptr(tail), 1024, 1, C.c_float(1.0 / (4.0 * w_scale)), stream_ptr())
        torch.cuda.synchronize()
        assert rc == 0 and torch.equal(y4, y) and torch.equal(a4, a_out)
    finally:
        lib.dic_debug_force_staged_gemm(113)


@pytest.mark.gpu
@pytest.mark.parametrize("B,Cin,CO,H", [(64, 256, 256, 14), (40, 96, 128, 14), (3, 64, 128, 14), (67, 512, 128, 14),
                                        (64, 128, 128, 28), (5, 64, 256, 28), (3, 32, 128, 28), (33, 128, 128, 28)])
def test_conv3x3_halo_kernel_with_on_the_fly_operand(lib, B, Cin, CO, H):
    """Round 4: the LDS-halo 3x3 kernel reading the RAW fp32 output of the convolution before it, its producer waves forming
    relu(raw * scale + shift), scaling by 4 and writing the two fp16 plane images (conv3x3_fwd_bf3_bn) - against the same kernel fed
    with planes of the same activation evaluated in torch (same LDS image, same products in the same order: bit-identical outputs -
    tests/test_encoders_gpu.py asserts exactly that against the bn_apply_planes route on the whole network), against fp64,
    BatchNorm partial sums, a ragged last tile / tiles spanning images / 2..16 channel chunks, repeated launches, and the overflow
    guard: one raw value whose activation leaves the fp16 range raises the status word (bit 4: producer waves).  H = 28: the same
    kernel with 9 halo rows of 32 padded pixels (ResNet layer 2), which exists in this form only."""
    import torch.nn.functional as F
    M = B * H * H
    g = torch.Generator().manual_seed(B + Cin + CO)
    raw = torch.randn(B, H, H, Cin, generator=g).to(DEV)
    scale = (torch.rand(Cin, generator=g) + 0.5).to(DEV)
    shift = (torch.randn(Cin, generator=g) * 0.3).to(DEV)
    w = (torch.randn(CO, 3, 3, Cin, generator=g) / (9 * Cin) ** 0.5).to(DEV)
    # the kernel's raw * scale + shift is ONE fused multiply-add; in fp64 the product of two floats is exact and the sum rounds at 53
    # bits, so rounding that to fp32 is the fused result (up to a double rounding in about one element in 2^29)
    act = torch.relu((raw.double() * scale.double() + shift.double()).float())
    w_scale = 2.0 ** math.floor(14 - math.log2(float(w.abs().max())))
    out_scale = C.c_float(1.0 / (4.0 * w_scale))

    def split(x2d, s):
        R, K = x2d.shape
        out = [torch.zeros((R + 1) // 2 * 2 * K, dtype=torch.int16, device=DEV) for _ in range(2)]
        check(lib.dic_split_f16x2_paired(ptr(x2d), C.c_longlong(R), K, C.c_float(s), ptr(out[0]), ptr(out[1]), stream_ptr()), "split")
        return out

    pl = lambda ps: (C.c_void_p * 3)(ps[0].data_ptr(), ps[1].data_ptr(), None)            # noqa: E731
    xp, wp = split(act.reshape(M, Cin).contiguous(), 4.0), split(w.view(CO, -1), w_scale)
    tail = torch.empty(1024 * 64 * 64, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    try:
        assert lib.dic_debug_force_staged_gemm(74) == 0            # the halo kernel whatever the tile count
        y_pl = torch.full((M, CO), float("nan"), device=DEV)
        check(lib.dic_debug_conv_fmt(pl(xp), B, H, H, Cin, pl(wp), CO, 3, 1, 1, ptr(y_pl), None, None, ptr(tail), 1, out_scale, stream_ptr()), "planes")
        ys = []
        for rep in range(3):
            y = torch.full((M, CO), float("nan"), device=DEV)
            part = torch.zeros((M // 64 + 2) * 2 * CO, device=DEV)
            mt = C.c_int(0)
            rc = lib.dic_debug_conv3x3_bn(ptr(raw), ptr(scale), ptr(shift), 1, B, H, H, Cin, pl(wp), CO, ptr(y), ptr(part), C.byref(mt), ptr(tail), 1024,
                                          out_scale, ptr(status), stream_ptr())
            assert rc == 0, (rc, lib.dic_last_error())
            torch.cuda.synchronize()
            ys.append(y)
        assert torch.isfinite(ys[0]).all() and int(status.item()) == 0
        assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2]), "repeated launches differ"
        # switch 120: the computing waves read their fragments in a block in front of each k-step's MFMAs (the form until round 4);
        # 121 (default): one read per MFMA gap.  Same products in the same order: outputs and BatchNorm partials bit for bit
        part_ilv = part.clone()
        try:
            assert lib.dic_debug_force_staged_gemm(120) == 0
            y_blk = torch.full((M, CO), float("nan"), device=DEV)
            part_blk = torch.zeros_like(part)
            rc = lib.dic_debug_conv3x3_bn(ptr(raw), ptr(scale), ptr(shift), 1, B, H, H, Cin, pl(wp), CO, ptr(y_blk), ptr(part_blk), C.byref(mt), ptr(tail),
                                          1024, out_scale, ptr(status), stream_ptr())
            torch.cuda.synchronize()
            assert rc == 0 and torch.equal(y_blk, ys[0]) and torch.equal(part_blk, part_ilv), "interleaved fragment reads changed the result"
        finally:
            lib.dic_debug_force_staged_gemm(121)
        ndiff = int((ys[0] != y_pl).sum())          # (a double-rounded activation element would show in its 9 * CO outputs, at rounding level)
        print(f"\noutputs that differ from the plane route: {ndiff} of {M * CO}")
        if H == 14:
            assert ndiff <= 3 * 9 * CO and float((ys[0] - y_pl).abs().max()) <= 2e-6 * float(y_pl.abs().max()), (ndiff, float((ys[0] - y_pl).abs().max()))
        else:          # 28x28 planes run on the gathered kernel (tap-major summation): same products, another order
            assert float((ys[0] - y_pl).abs().max()) <= 4e-6 * float(y_pl.abs().max())
        ref = F.conv2d(act.permute(0, 3, 1, 2).double().cpu(), w.permute(0, 3, 1, 2).double().cpu(), padding=1).permute(0, 2, 3, 1).reshape(M, CO)
        sc = float(ref.abs().max())
        err = float((ys[0].double().cpu() - ref).abs().max()) / sc
        print(f"\n3x3 halo, on-the-fly operand, {B}x{H}x{H}x{Cin} -> {CO}: max err / scale vs fp64 {err:.2e}")
        assert err < 4e-6
        stats = part[: mt.value * 2 * CO].view(mt.value, 2, CO).double().sum(0).cpu()
        assert torch.allclose(stats[0], ref.sum(0), rtol=1e-4, atol=1e-3 * sc) and torch.allclose(stats[1], (ref * ref).sum(0), rtol=1e-4, atol=1e-3 * sc)
        # the guard: relu(2e4 * scale + shift) * 4 is beyond fp16
        raw2 = raw.clone()
        raw2[B // 2, 5, 7, 3] = 4.0e4
        rc = lib.dic_debug_conv3x3_bn(ptr(raw2), ptr(scale), ptr(shift), 1, B, H, H, Cin, pl(wp), CO, ptr(y), ptr(part), C.byref(mt), ptr(tail), 1024,
                                      out_scale, ptr(status), stream_ptr())
        torch.cuda.synchronize()
        assert rc == 0 and int(status.item()) & 4, int(status.item())
    finally:
        lib.dic_debug_force_staged_gemm(78)
    # a shape the halo kernel does not take: nothing launched, the caller's cue to take the plane route
    y2 = torch.zeros(56 * 56, CO, device=DEV)
    assert lib.dic_debug_conv3x3_bn(ptr(raw), ptr(scale), ptr(shift), 1, 1, 56, 56, Cin, pl(wp), CO, ptr(y2), None, None, ptr(tail), 1024, out_scale,
                                    ptr(status), stream_ptr()) == 1
    torch.cuda.synchronize()
    assert not bool(y2.any())


@pytest.mark.gpu
def test_f16x2_activation_beyond_fp16_range_fails_loudly(lib):
    """The f16x2 format stores 4 * x in fp16: an activation beyond +-16376 cannot be represented.  The documented behaviour is a loud
    one - the plane holds inf, every output that touches it is inf / NaN - never a silently clamped value; bf16x3 on the same input
    is exact."""
    M, Cin, CO = 12544, 256, 256
    g = torch.Generator().manual_seed(3)
    x = torch.randn(M, Cin, generator=g).to(DEV)
    x[7, 5] = 2.0e4
    w = (torch.randn(CO, Cin, generator=g) / Cin ** 0.5).to(DEV)
    w_scale = 2.0 ** math.floor(14 - math.log2(float(w.abs().max())))
    tail = torch.empty(1024 * 64 * 64, device=DEV)
    outs = {}
    for fmt in (0, 1):
        xp = [torch.zeros(M * Cin, dtype=torch.int16, device=DEV) for _ in range(3 - fmt)] + ([None] if fmt else [])
        wp = [torch.zeros(CO * Cin, dtype=torch.int16, device=DEV) for _ in range(3 - fmt)] + ([None] if fmt else [])
        if fmt:
            check(lib.dic_split_f16x2_paired(ptr(x), C.c_longlong(M), Cin, C.c_float(4.0), ptr(xp[0]), ptr(xp[1]), stream_ptr()), "split x")
            check(lib.dic_split_f16x2_paired(ptr(w), C.c_longlong(CO), Cin, C.c_float(w_scale), ptr(wp[0]), ptr(wp[1]), stream_ptr()), "split w")
        else:
            check(lib.dic_split_bf16x3_paired(ptr(x), C.c_longlong(M), Cin, ptr(xp[0]), ptr(xp[1]), ptr(xp[2]), stream_ptr()), "split x")
            check(lib.dic_split_bf16x3_paired(ptr(w), C.c_longlong(CO), Cin, ptr(wp[0]), ptr(wp[1]), ptr(wp[2]), stream_ptr()), "split w")
        y = torch.zeros(M, CO, device=DEV)
        pl = lambda ps: (C.c_void_p * 3)(*[t.data_ptr() if t is not None else None for t in ps])            # noqa: E731
        check(lib.dic_debug_conv_fmt(pl(xp), 1, 1, M, Cin, pl(wp), CO, 1, 1, 0, ptr(y), None, None, ptr(tail), fmt,
                                     C.c_float(1.0 / (4.0 * w_scale)), stream_ptr()), "conv")
        torch.cuda.synchronize()
        outs[fmt] = y
    assert torch.isfinite(outs[0]).all()
    bad_rows = (~torch.isfinite(outs[1])).any(1).nonzero().flatten().tolist()
    assert bad_rows == [7], bad_rows          # exactly the output row that consumed the out-of-range activation, nothing silent elsewhere
    ok = torch.ones(M, dtype=torch.bool); ok[7] = False
    ref = (x.double().cpu() @ w.double().cpu().t())
    assert float((outs[1].double().cpu()[ok] - ref[ok]).abs().max()) < 4e-6 * float(ref[ok].abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("rows,K", [(64, 256), (37, 96)])
def test_split_f16x2_planes_equal_the_numpy_statement(lib, rows, K):
    """dic_split_f16x2_paired against tests/test_f16x2_cpu.py::split2_f16 (numpy, IEEE round-to-nearest-even), bit for bit, in the
    row-pair interleaved layout (include/dic.h) - values from the top of the fp16 range down into the subnormal second plane; an odd
    row count leaves a zero pad row."""
    import numpy as np
    from test_f16x2_cpu import split2_f16
    rng = np.random.default_rng(rows)
    x = (rng.standard_normal((rows, K)) * np.exp(rng.uniform(-18, 8.0, (rows, K)))).astype(np.float32)
    x[np.abs(x) >= 16000] = 1.0
    xt = torch.from_numpy(x).to(DEV)
    n = (rows + 1) // 2 * 2 * K
    h1 = torch.full((n,), 0x7fff, dtype=torch.int16, device=DEV)
    h2 = torch.full((n,), 0x7fff, dtype=torch.int16, device=DEV)
    check(lib.dic_split_f16x2_paired(ptr(xt), C.c_longlong(rows), K, C.c_float(4.0), ptr(h1), ptr(h2), stream_ptr()), "split")
    torch.cuda.synchronize()
    e1, e2 = split2_f16(x, 4.0)
    pad = (rows + 1) // 2 * 2
    want = []
    for e in (e1, e2):
        full = np.zeros((pad, K), np.float16)
        full[:rows] = e
        # element (r, k) lives at ((r/2)*(K/32) + k/32)*64 + (r%2)*32 + k%32
        want.append(full.reshape(pad // 2, 2, K // 32, 32).transpose(0, 2, 1, 3).reshape(-1).view(np.int16))
    assert np.array_equal(h1.cpu().numpy(), want[0]), "first plane"
    assert np.array_equal(h2.cpu().numpy(), want[1]), "second plane"

