"""GPU tests of the PARKED kernels (built, measured, not used by the product: DESIGN.md 5.1, 5.3, 10).  They live in
libdic_experiments.so only (python -m depth_image_captioning_pub_amd.build --experiments) and are not part of `pytest tests/`:

    DIC_LIB=experiments python -m pytest scripts/experiments -q
"""
import ctypes as C
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
# (no import-time change of the environment: the library is chosen by the DIC_LIB the caller exports - a collected but
#  deselected copy of this file must not redirect the product suite to libdic_experiments.so)

from depth_image_captioning_pub_amd import _lib, native, synthetic as syn      # noqa: E402
from depth_image_captioning_pub_amd._lib import check, ptr, stream_ptr          # noqa: E402
from oracle import captioning_oracle as orc                                      # noqa: E402
from tests.test_decoder_gpu import _assert_close, _inputs, _to_dev              # noqa: E402
import math                                                                      # noqa: E402

DEV = "cuda:0"
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("DIC_LIB") != "experiments" or not torch.cuda.is_available() or
                                 not os.path.exists(_lib.LIB_EXPERIMENTS_PATH),
                                 reason="run as `DIC_LIB=experiments python -m pytest scripts/experiments` on a GPU box with "
                                        "libdic_experiments.so built")]


@pytest.fixture(scope="module")
def lib():
    assert os.environ.get("DIC_LIB") == "experiments"
    return _lib.load()


@pytest.mark.parametrize("cells", [49, 196])
@pytest.mark.parametrize("lengths", [[21] * 64, [13, 13, 12, 9, 9, 8, 5, 2], [7] * 5])
def test_persistent_forward_loop_vs_oracle_and_per_step(lib, lengths, cells):
    """The opt-in persistent forward loop (csrc/experiments/decoder_persist.hip: one launch for all T steps, 256 co-resident
    workgroups exchanging partial gate pre-activations through write-through stores + per-group counters) against the oracle
    (logits / alphas 1e-4, argmax identical) and against the default per-step launches (1e-5: only the summation order of
    the gate pre-activation differs), full (64 rows), ragged and partial-group batches, both layouts; the hand-off status word
    must stay clear and the backward must accept the tape it leaves."""
    vocab, seed = 300, 91
    w, f_rgb, f_dep, caps, lens = _inputs(lengths, vocab, seed, replicate=True)
    B, tmax = len(lens), max(lens) - 1
    drop = syn.dropout_multiplier(B, tmax, 0.5, seed=seed)
    ref, _, al_ref = orc.decoder_forward(w, f_rgb, f_dep, caps, lens, drop)

    def cut(f):
        return f if cells == 196 else f.reshape(B, 14, 14, -1)[:, ::2, ::2].reshape(B, 49, -1).contiguous()
    out = {}
    try:
        for name, code in (("per_step", 140), ("persistent", 141)):
            lib.dic_debug_force_staged_gemm(code)
            logits, alphas, tape = native.decoder_forward(_to_dev(w), cut(f_rgb).to(DEV), cut(f_dep).to(DEV), caps.to(DEV), lens,
                                                          drop.to(DEV))
            loss, dl, da = native.caption_loss(logits, native.pack_targets(caps.to(DEV), lens), alphas)
            grads, _ = native.decoder_backward(tape, dl, da)
            torch.cuda.synchronize()
            out[name] = (logits.clone(), alphas.clone(), float(loss.item()), {k: v.clone() for k, v in grads.items()})
    finally:
        lib.dic_debug_force_staged_gemm(140)
    _assert_close("logits", out["persistent"][0], ref, 1e-4)
    _assert_close("alphas", out["persistent"][1], al_ref, 1e-4)
    assert torch.equal(out["persistent"][0].argmax(1).cpu(), ref.argmax(1))
    _assert_close("logits vs per-step", out["persistent"][0], out["per_step"][0], 1e-5)
    assert abs(out["persistent"][2] - out["per_step"][2]) <= 1e-5
    for k in w:
        if not k.endswith("full_att.bias"):
            _assert_close("grad." + k, out["persistent"][3][k], out["per_step"][3][k], 1e-4)



@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (1000, 300, 96), (129, 130, 64), (40000, 384, 160), (33000, 256, 64),
                                   (4096, 1024, 1056)])
def test_parked_tile_variants_are_bit_identical(lib, M, N, K):
    """The parked forms of the split-bf16 kernel - deep-pipelined 128x128 (23), persistent with DMA issued by the computing
    waves (24 under 77), 256x128 warp-specialised (26) - against the product's 64x64 kernel (11): bit for bit."""
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
        for code in (11, 23, 2477, 26):
            lib.dic_debug_force_staged_gemm(77 if code == 2477 else 76)
            lib.dic_debug_force_staged_gemm(24 if code == 2477 else code)
            for rep in range(3):
                Cm = torch.full((M, N), float("nan"), device=DEV)
                check(lib.dic_gemm_bf16x3_paired(M, N, K, ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(b[0]), ptr(b[1]), ptr(b[2]),
                                                 ptr(Cm), C.c_longlong(N), None, stream_ptr()), "dic_gemm_bf16x3_paired")
                assert torch.isfinite(Cm).all(), code
                if code in outs:
                    assert torch.equal(outs[code], Cm), f"code {code}: repetition {rep} differs"
                outs[code] = Cm
    finally:
        lib.dic_debug_force_staged_gemm(76)
        lib.dic_debug_force_staged_gemm(20)
    for code in (23, 2477, 26):
        assert torch.equal(outs[11], outs[code]), f"parked variant {code} differs from the 64x64 kernel"


@pytest.mark.parametrize("M,Cin,CO", [(12544, 256, 1024), (50176, 128, 512), (12500, 256, 1024), (392, 256, 1024), (6272, 128, 512), (64, 256, 128)])
def test_conv1x1_a_stationary_kernel(lib, M, Cin, CO):
    """conv1x1_astat_bn_kernel (round 4; ResNet conv3 of layers 2 and 3 in the f16x2 format): y = relu(raw * scale + shift) . W^T with the
    64-row input block normalised / rectified / split ONCE inside the kernel and resident in LDS for all output columns.  Against fp64,
    against the plane route on the same kernel family (dic_split_f16x2_paired of the torch-evaluated activation -> dic_debug_conv_fmt;
    printed: whether the outputs are bit-identical), BatchNorm partial sums (per 32-row wave tile) against the stored output; ragged
    last row blocks (12500, 392) leave rows past M untouched."""
    assert lib.dic_debug_force_staged_gemm(111) == 0          # the parked kernel is off unless asked for
    g = torch.Generator().manual_seed(M + Cin)
    raw = torch.randn(M, Cin, generator=g).to(DEV)
    scale = (torch.rand(Cin, generator=g) + 0.5).to(DEV)
    shift = (torch.randn(Cin, generator=g) * 0.3).to(DEV)
    w = (torch.randn(CO, Cin, generator=g) / Cin ** 0.5).to(DEV)
    act = torch.relu(torch.addcmul(shift, raw, scale))
    w_scale = 2.0 ** math.floor(14 - math.log2(float(w.abs().max())))
    wp = [torch.zeros((CO + 1) // 2 * 2 * Cin, dtype=torch.int16, device=DEV) for _ in range(2)]
    check(lib.dic_split_f16x2_paired(ptr(w), C.c_longlong(CO), Cin, C.c_float(w_scale), ptr(wp[0]), ptr(wp[1]), stream_ptr()), "split w")
    wpl = (C.c_void_p * 3)(wp[0].data_ptr(), wp[1].data_ptr(), None)
    pad = 3                                                        # canary rows behind the matrix
    y = torch.full((M + pad, CO), float("nan"), device=DEV)
    rows = 2 * ((M + 63) // 64)
    part = torch.full((rows * 2 * CO,), float("nan"), device=DEV)
    mt = C.c_int(0)
    word = torch.zeros(1, dtype=torch.int32, device=DEV)
    rc = lib.dic_debug_conv1x1_astat(ptr(raw), ptr(scale), ptr(shift), 1, M, Cin, wpl, CO, ptr(y), ptr(part), C.byref(mt),
                                     C.c_float(1.0 / (4.0 * w_scale)), ptr(word), stream_ptr())
    assert rc == 0, (rc, lib.dic_last_error())
    torch.cuda.synchronize()
    assert mt.value == rows and int(word.item()) == 0
    assert torch.isnan(y[M:]).all(), "rows past M were written"
    y = y[:M]
    assert torch.isfinite(y).all()
    ref64 = act.double().cpu() @ w.double().cpu().t()
    sc = float(ref64.abs().max())
    err = float((y.double().cpu() - ref64).abs().max()) / sc
    yd = y.double().cpu()
    stats = part.view(rows, 2, CO).double().sum(0).cpu()
    assert torch.allclose(stats[0], yd.sum(0), rtol=1e-5, atol=1e-4 * sc) and torch.allclose(stats[1], (yd * yd).sum(0), rtol=1e-5, atol=1e-4 * sc)
    # the plane route on the same values
    xp = [torch.zeros((M + 1) // 2 * 2 * Cin, dtype=torch.int16, device=DEV) for _ in range(2)]
    check(lib.dic_split_f16x2_paired(ptr(act), C.c_longlong(M), Cin, C.c_float(4.0), ptr(xp[0]), ptr(xp[1]), stream_ptr()), "split x")
    y2 = torch.zeros(M, CO, device=DEV)
    tail = torch.empty(1024 * 64 * 64, device=DEV)
    xpl = (C.c_void_p * 3)(xp[0].data_ptr(), xp[1].data_ptr(), None)
    check(lib.dic_debug_conv_fmt(xpl, 1, 1, M, Cin, wpl, CO, 1, 1, 0, ptr(y2), None, None, ptr(tail), 1, C.c_float(1.0 / (4.0 * w_scale)),
                                 stream_ptr()), "plane route")
    torch.cuda.synchronize()
    err2 = float((y2.double().cpu() - ref64).abs().max()) / sc
    print(f"\n{M}x{CO}x{Cin} A-stationary: max err / scale vs fp64 {err:.2e} (plane route {err2:.2e}); bit-identical to the plane route: "
          f"{bool(torch.equal(y, y2))}")
    assert err < 4e-6 and err <= 2.0 * err2 + 1e-6
    # a value beyond the fp16 range of the planes raises the guard word; shapes that are not this kernel's are refused (nothing launched)
    raw2 = raw.clone(); raw2[5, 7] = 1.0e6
    word.zero_()
    assert lib.dic_debug_conv1x1_astat(ptr(raw2), ptr(scale), ptr(shift), 1, M, Cin, wpl, CO, ptr(y), ptr(part), C.byref(mt),
                                       C.c_float(1.0 / (4.0 * w_scale)), ptr(word), stream_ptr()) == 0
    torch.cuda.synchronize()
    assert int(word.item()) != 0
    assert lib.dic_debug_conv1x1_astat(ptr(raw), ptr(scale), ptr(shift), 1, M, 64, wpl, CO, ptr(y), None, None, C.c_float(1.0), None,
                                       stream_ptr()) == 1
    lib.dic_debug_force_staged_gemm(110)


def test_resnet_forward_conv3_on_the_a_stationary_kernel_matches_plane_route(lib):
    """Round 4: in the f16x2 format conv3 of ResNet layers 2 and 3 runs on conv1x1_astat_bn_kernel (switch 111, default) instead of
    bn_apply_planes + the twelve-wave plane kernel (110).  Same operand values, same products, same summation order per output element -
    what differs is the granularity of the BatchNorm partial sums (32-row instead of 64-row wave tiles: another association of the
    same fp32 column sums), i.e. rounding level, amplified through the following BatchNorm layers like any reordering: the feature map
    agrees to 2e-3 of its scale (the bound the folding test above uses for re-associated sums), running statistics to 1e-4, and each
    route reproduces itself bit for bit."""
    w = syn.resnet152_weights(seed=125)
    imgs = syn.rgb_images(64, seed=123).to(DEV)
    results = {}
    try:
        for code in (110, 111, 111, 110):
            assert lib.dic_debug_force_staged_gemm(code) == 0
            wd = {k: v.to(DEV) for k, v in w.items()}
            y = native.ResNetRunner(wd, conv_mode="f16x2").forward(imgs, train_bn=True, compact=True)
            torch.cuda.synchronize()
            assert torch.isfinite(y).all()
            stats = torch.cat([wd[k].flatten() for k in sorted(wd) if "running" in k])
            if code in results:
                assert torch.equal(results[code][0], y) and torch.equal(results[code][1], stats), f"switch {code} does not reproduce itself"
            results[code] = (y.clone(), stats.clone())
    finally:
        lib.dic_debug_force_staged_gemm(110)
    (y0, s0), (y1, s1) = results[110], results[111]
    dy, ds = float((y1 - y0).abs().max()) / float(y0.abs().max()), float((s1 - s0).abs().max()) / float(s0.abs().max())
    print(f"\nconv3 A-stationary vs plane route: features max |d| / max = {dy:.2e}, running statistics {ds:.2e}")
    assert dy < 2e-3 and ds < 1e-4, (dy, ds)


@pytest.mark.parametrize("M,Cin,CO", [(12544, 256, 1024), (50176, 128, 512), (3136, 512, 2048), (12500, 256, 1024), (12544, 256, 256)])
def test_conv1x1_on_the_fly_operand_on_the_256x128_kernel(lib, M, Cin, CO):
    """Round 4: conv3 reading conv2's RAW output - relu(raw * scale + shift), no residual, no fp32 copy - on the twelve-wave 256x128
    kernel, its four producer waves forming the two fp16 plane images (switch 107; PARKED: neutral in the step) instead of the 128x128 kernel's (106):
    same element-wise arithmetic, same products in the same order as the same kernel fed with planes -> bit-identical to that route
    (the 128x128 kernel may cut its remainder tiles into K slices: rounding level), BatchNorm partial sums of the same totals; fp64; a ragged last tile (12500 rows), the guard, and a shape the policy keeps on the 128x128 kernel
    (256 output channels: too few 256x128 tiles)."""
    g = torch.Generator().manual_seed(M + Cin + CO)
    raw = torch.randn(M, Cin, generator=g).to(DEV)
    scale = (torch.rand(Cin, generator=g) + 0.5).to(DEV)
    shift = (torch.randn(Cin, generator=g) * 0.3).to(DEV)
    w = (torch.randn(CO, Cin, generator=g) / Cin ** 0.5).to(DEV)
    act = torch.relu((raw.double() * scale.double() + shift.double()).float())
    w_scale = 2.0 ** math.floor(14 - math.log2(float(w.abs().max())))
    wp = [torch.zeros((CO + 1) // 2 * 2 * Cin, dtype=torch.int16, device=DEV) for _ in range(2)]
    check(lib.dic_split_f16x2_paired(ptr(w), C.c_longlong(CO), Cin, C.c_float(w_scale), ptr(wp[0]), ptr(wp[1]), stream_ptr()), "split w")
    wpl = (C.c_void_p * 3)(wp[0].data_ptr(), wp[1].data_ptr(), None)
    tail = torch.empty(1024 * 64 * 64, device=DEV)
    out = {}
    try:
        for code in (106, 107, 107):
            assert lib.dic_debug_force_staged_gemm(code) == 0
            y = torch.full((M, CO), float("nan"), device=DEV)
            part = torch.zeros((M // 64 + 2) * 2 * CO, device=DEV)
            mt = C.c_int(0)
            rc = lib.dic_debug_conv1x1_bn_fmt(ptr(raw), ptr(scale), ptr(shift), None, 1, None, M, Cin, wpl, CO, ptr(y), ptr(part), C.byref(mt),
                                              ptr(tail), 1024, 1, C.c_float(1.0 / (4.0 * w_scale)), stream_ptr())
            assert rc == 0, (rc, lib.dic_last_error())
            torch.cuda.synchronize()
            assert torch.isfinite(y).all()
            st = part[: mt.value * 2 * CO].view(mt.value, 2, CO).double().sum(0).cpu()
            if code in out:
                assert torch.equal(out[code][0], y), "repeated launch differs"
            out[code] = (y, st)
        # the plane route: the same 256x128 kernel fed with planes of the activation (same LDS image, same products in the same order)
        xp = [torch.zeros((M + 1) // 2 * 2 * Cin, dtype=torch.int16, device=DEV) for _ in range(2)]
        check(lib.dic_split_f16x2_paired(ptr(act), C.c_longlong(M), Cin, C.c_float(4.0), ptr(xp[0]), ptr(xp[1]), stream_ptr()), "split x")
        y_pl = torch.full((M, CO), float("nan"), device=DEV)
        check(lib.dic_debug_conv_fmt((C.c_void_p * 3)(xp[0].data_ptr(), xp[1].data_ptr(), None), 1, 1, M, Cin, wpl, CO, 1, 1, 0, ptr(y_pl), None, None,
                                     ptr(tail), 1, C.c_float(1.0 / (4.0 * w_scale)), stream_ptr()), "planes")
        torch.cuda.synchronize()
    finally:
        lib.dic_debug_force_staged_gemm(106)
    ndiff = int((out[107][0] != y_pl).sum())         # (a double-rounded activation element of the torch evaluation would show in its CO outputs)
    print(f"\noutputs that differ from the plane route: {ndiff} of {M * CO}")
    assert ndiff <= 3 * CO and float((out[107][0] - y_pl).abs().max()) <= 2e-6 * float(y_pl.abs().max())
    # the 128x128 kernel takes some of these shapes with the remainder-round K split: another association of the same products
    assert float((out[106][0] - out[107][0]).abs().max()) <= 4e-6 * float(y_pl.abs().max())
    ref64 = act.double().cpu() @ w.double().cpu().t()
    sc = float(ref64.abs().max())
    err = float((out[107][0].double().cpu() - ref64).abs().max()) / sc
    print(f"\n{M}x{CO}x{Cin} on-the-fly operand, 256x128 kernel: max err / scale vs fp64 {err:.2e}")
    assert err < 4e-6
    for code in (106, 107):      # the partial sums are those of the stored fp32 outputs (per-tile fp32 sums: tolerance grows with the rows)
        yd, st = out[code][0].double().cpu(), out[code][1]
        tol = 1e-4 * sc * max(1.0, M / 12544)
        assert torch.allclose(st[0], yd.sum(0), rtol=1e-5, atol=tol) and torch.allclose(st[1], (yd * yd).sum(0), rtol=1e-5, atol=tol)


@pytest.mark.parametrize("B,Cin,CO", [(64, 64, 64), (3, 64, 64), (2, 32, 128), (5, 64, 64)])
def test_conv3x3_halo_kernel_for_56x56_maps(lib, B, Cin, CO):
    """Round 4, parked (switch 127): the LDS-halo 3x3 kernel with the on-the-fly operand in a 7-row x 58-pixel geometry for ResNet layer 1
    (64 -> 64 channels on half of the 128-column tile).  Same checks as the product's 14x14 / 28x28 forms - planes route at rounding
    level, fp64, BatchNorm partials, ragged tiles, repeated launches, the overflow guard, bit-identity of the two fragment-read
    schedules - by running the product test's body on this geometry.  Correct, and no faster than the planes pass + gathered kernel it
    would replace (110 us against 17 + 85; csrc/gemm_bf3.hip)."""
    from tests.test_gemm_gpu import test_conv3x3_halo_kernel_with_on_the_fly_operand as body
    try:
        assert lib.dic_debug_force_staged_gemm(127) == 0
        body(lib, B, Cin, CO, 56)
    finally:
        lib.dic_debug_force_staged_gemm(126)
