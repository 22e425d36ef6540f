"""GPU parity of the two CNN encoders against the CPU oracle (and the reference's golden vectors for
the depth encoder).  Tolerances: fp32 with different summation order through BatchNorm in batch-statistics
mode: 2e-4 of the output scale for the 3-conv depth encoder; the 152-layer ResNet (parity UNPINNED: no
torchvision in the build container, oracle = torch conv/batch_norm restatement) gets 2e-3."""
import numpy as np
import pytest
import torch

from depth_image_captioning_pub_amd import native, synthetic as syn
from oracle import captioning_oracle as orc
from tests.helpers import check_packed, load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(name, got, ref, tol, atol=0.0):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert got.shape == ref.shape, f"{name}: {tuple(got.shape)} vs {tuple(ref.shape)}"
    scale = float(ref.abs().max()) + 1e-12
    err = float((got - ref).abs().max())
    assert np.isfinite(err) and err <= tol * scale + atol, f"{name}: max err {err:.3e} > {tol:g}*{scale:.3e}+{atol:g}"


def _dev(d):
    return {k: v.to(DEV).contiguous() for k, v in d.items()}


# sizes: 224 = golden case (73-pixel rows: ragged last pixel group of 1); 100 -> 32 pixels/row (no ragged group); 109 -> 35
# (ragged 3); 300 / 520 exercise the wider row-prefetch instantiations of the layer-1 weight-gradient kernel (W <= 512,
# W <= 640).  Seeds of the small maps come from scripts/sweep_depth_seeds.py: with 18..288 samples per channel in layer 3 a
# ReLU / max-pool decision within ~1e-6 of a tie flips between ANY two fp32 summation orders and moves gradients by
# percents (seen for the old and the new layer-1 kernels alike); these seeds have no such near-tie.
@pytest.mark.parametrize("B,size,seed", [(2, 224, 51), (3, 100, 52), (2, 109, 54), (1, 300, 54), (1, 520, 54)])
def test_depth_encoder_train_fwd_bwd(lib, B, size, seed):
    w, st = syn.depth_encoder_weights(seed=seed)
    # non-trivial BN affine parameters so dgamma/dbeta paths are exercised
    g = torch.Generator().manual_seed(seed)
    for i in (1, 2, 3):
        w[f"bn{i}.weight"] = 1.0 + 0.2 * torch.randn(w[f"bn{i}.weight"].shape, generator=g)
        w[f"bn{i}.bias"] = 0.1 * torch.randn(w[f"bn{i}.bias"].shape, generator=g)
    if size == 224:                       # keep the golden configuration exact
        w, st = syn.depth_encoder_weights(seed=seed)
    depth = syn.depth_maps(B, seed=seed, size=size)
    d_out = torch.from_numpy(np.random.Generator(np.random.PCG64(seed + 7)).standard_normal((B, 196, 2048))
                             .astype(np.float32)) * 1e-2
    st_ref = {k: v.clone() for k, v in st.items()}
    wg = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    y_ref = orc.depth_encoder_forward(wg, st_ref, depth, train=True)
    (y_ref * d_out).sum().backward()
    st_dev = _dev(st)
    y, tape = native.depth_encoder_forward(_dev(w), st_dev, depth.to(DEV), train=True)
    _close("features", y, y_ref, 2e-4)
    for k in st:
        _close(k, st_dev[k], st_ref[k], 1e-4)
    grads = native.depth_encoder_backward(tape, d_out.to(DEV))
    try:
        for k in w:
            # conv biases feed train-mode BN: true gradient 0, only rounding noise on both sides
            if k.startswith("conv") and k.endswith("bias"):
                _close("grad." + k, grads[k], wg[k].grad, 0.0, atol=5e-5)
            else:
                _close("grad." + k, grads[k], wg[k].grad, 2e-3)
    except AssertionError as direct:
        # A direct comparison of gradients with the fp32 oracle holds only while both sides take the same ReLU / max-pool
        # selections; a selection that is a tie at fp32 rounding level may fall either way (B = 1 at 520x520 did once the
        # convolution's summation order changed: one flipped tie moved grad.conv2.weight by 1 % of its scale).  The stricter
        # statement then has to hold: with the HIP path's selections replayed, everything agrees with fp64 to 2e-4 and every
        # differing selection is a tie (the check the *_decision_replay tests run on all seeds).
        if size == 224:
            raise
        print("direct gradient comparison failed (", str(direct).split(chr(10))[0], ") - checking with replayed selections")
        _replay_check(w, st, depth, d_out)
    if size == 224:
        gold = load_golden("depth_encoder_train")
        check_packed(gold, "out49", y.reshape(B, 14, 14, 2048)[:, ::2, ::2].reshape(B, 49, 2048), 1e-3, 1e-4)
        for k in w:
            if k.startswith("conv") and k.endswith("bias"):
                continue
            gk = grads[k].cpu()
            check_packed(gold, "grad." + k, gk, 5e-3, 2e-3 * float(gk.abs().max()))


def _replay_check(w, st, depth, d_out, tie_tol=3e-5, grad_tol=2e-4):
    """HIP forward/backward of the depth encoder vs an fp64 evaluation of the oracle that REPLAYS the HIP path's ReLU /
    max-pool selections (dic_depth_encoder_inspect -> orc.depth_encoder_forward_replay).  Returns the tie report."""
    st_dev = _dev(st)
    y, tape = native.depth_encoder_forward(_dev(w), st_dev, depth.to(DEV), train=True)
    grads = native.depth_encoder_backward(tape, d_out.to(DEV))
    dec = {k: v.cpu() for k, v in native.depth_encoder_decisions(tape).items()}
    w64 = {k: v.double().clone().requires_grad_(True) for k, v in w.items()}
    y64, rep = orc.depth_encoder_forward_replay(w64, {k: v.double().clone() for k, v in st.items()}, depth.double(), dec)
    (y64 * d_out.double()).sum().backward()
    # 1. every selection that differs from the fp64 evaluation's own choice is a tie-break at fp32 rounding level
    for name, (count, shortfall) in rep.items():
        assert shortfall <= tie_tol, f"{name}: {count} selections differ from fp64, worst shortfall {shortfall:.2e} of the map's scale"
    # 2. with the selections fixed, outputs and all gradients agree with fp64 tightly - no seed is special
    _close("features", y, y64.detach(), 2e-5)
    for k in w:
        if k.startswith("conv") and k.endswith("bias"):       # quirk Q10: exactly zero true gradient; what both sides hold
            # is the rounding noise of a sum over B*H*W terms - bounded relative to the sibling weight gradient's scale
            _close("grad." + k, grads[k], w64[k].grad, 0.0, atol=1e-5 * float(w64[k[:-4] + "weight"].grad.abs().max()) + 1e-7)
        else:
            _close("grad." + k, grads[k], w64[k].grad, grad_tol)
    return rep


@pytest.mark.parametrize("size", [300, 520])
def test_depth_encoder_all_seeds_with_fp64_decision_replay(lib, size):
    """Seeds 52..63 at the sizes where a plain comparison with the fp32 oracle fails for about half of the seeds
    (gpurun_out/seed_sweep.log of round 1; scripts/diag_depth_encoder_fp64.py shows the failures are symmetric: on some
    seeds the fp32 ORACLE is the one that sits percents away from fp64).  Cause: ReLU / max-pool selections within fp32
    rounding of a tie.  With the HIP path's selections replayed in an fp64 evaluation of the oracle, every seed matches at
    2e-4 of each gradient's scale, and every differing selection is shown to be a tie-break (shortfall <= 3e-5)."""
    flips = 0
    for seed in range(52, 64):
        w, st = syn.depth_encoder_weights(seed=seed)
        g = torch.Generator().manual_seed(seed)
        for i in (1, 2, 3):
            w[f"bn{i}.weight"] = 1.0 + 0.2 * torch.randn(w[f"bn{i}.weight"].shape, generator=g)
            w[f"bn{i}.bias"] = 0.1 * torch.randn(w[f"bn{i}.bias"].shape, generator=g)
        depth = syn.depth_maps(1, seed=seed, size=size)
        d_out = torch.from_numpy(np.random.Generator(np.random.PCG64(seed + 7)).standard_normal((1, 196, 2048))
                                 .astype(np.float32)) * 1e-2
        rep = _replay_check(w, st, depth, d_out)
        flips += sum(c for c, _ in rep.values())
    print(f"size {size}: {flips} tie-break selections differing from fp64 over 12 seeds")


@pytest.mark.parametrize("B", [16, 64])
def test_depth_encoder_bench_shape_with_fp64_decision_replay(lib, B):
    """Depth-encoder forward + backward at the bench shape (224x224, batch 16 and 64) against the fp64 decision-replay
    oracle: outputs 2e-5, all 12 gradients 2e-4 of their scale."""
    w, st = syn.depth_encoder_weights(seed=124)
    depth = syn.depth_maps(B, seed=123)
    d_out = torch.from_numpy(np.random.Generator(np.random.PCG64(131)).standard_normal((B, 196, 2048))
                             .astype(np.float32)) * 1e-2
    _replay_check(w, st, depth, d_out)


def test_depth_encoder_wide_map_uses_generic_layer1(lib):
    """A wide, flat map (52 x 700; wider than the 640-float rows the packed-FMA layer-1 kernels stage in LDS) falls back to the
    generic gather kernels (same results, same API): non-square feature grid pooled to 14 x 14."""
    w, st = syn.depth_encoder_weights(seed=61)
    g = torch.Generator().manual_seed(61)
    depth = torch.rand((1, 1, 52, 700), generator=g)
    d_out = torch.randn((1, 196, 2048), generator=g) * 1e-2
    st_ref = {k: v.clone() for k, v in st.items()}
    wg = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    y_ref = orc.depth_encoder_forward(wg, st_ref, depth, train=True)
    (y_ref * d_out).sum().backward()
    st_dev = _dev(st)
    y, tape = native.depth_encoder_forward(_dev(w), st_dev, depth.to(DEV), train=True)
    _close("features", y, y_ref, 2e-4)
    for k in st:
        _close(k, st_dev[k], st_ref[k], 1e-4)
    grads = native.depth_encoder_backward(tape, d_out.to(DEV))
    assert all(bool(torch.isfinite(v).all()) for v in grads.values())
    _close("grad.conv1.weight", grads["conv1.weight"], wg["conv1.weight"].grad, 5e-3)


def test_depth_encoder_eval_mode(lib):
    w, st = syn.depth_encoder_weights(seed=51)
    g = torch.Generator().manual_seed(1)
    for i, c in ((1, 128), (2, 512), (3, 2048)):
        st[f"bn{i}.running_mean"] = 0.1 * torch.randn(c, generator=g)
        st[f"bn{i}.running_var"] = 0.5 + torch.rand(c, generator=g)
    depth = syn.depth_maps(2, seed=51)
    y_ref = orc.depth_encoder_forward(w, {k: v.clone() for k, v in st.items()}, depth, train=False)
    st_dev = _dev(st)
    y, _ = native.depth_encoder_forward(_dev(w), st_dev, depth.to(DEV), train=False)
    _close("features", y, y_ref, 2e-4)
    for k in st:
        assert torch.equal(st_dev[k].cpu(), st[k]), "eval mode must not touch the running statistics"


@pytest.mark.parametrize("mode", ["fp32", "bf16x3", "f16x2"])
@pytest.mark.parametrize("layers,B,size,train", [((1, 1, 1, 1), 2, 64, True), ((1, 1, 1, 1), 2, 64, False),
                                                  ((2, 1, 2, 1), 3, 96, True)])
def test_resnet_small_stacks(lib, layers, B, size, train, mode):
    w = syn.resnet152_weights(seed=125, layers=layers)
    g = torch.Generator().manual_seed(7)
    for k in list(w):
        if k.endswith("running_mean"):
            w[k] = 0.05 * torch.randn(w[k].shape, generator=g)
        elif k.endswith("running_var"):
            w[k] = 0.8 + 0.4 * torch.rand(w[k].shape, generator=g)
        elif k.endswith(".bias"):
            w[k] = 0.1 * torch.randn(w[k].shape, generator=g)
    x = syn.rgb_images(B, seed=5, size=size)
    w_ref = {k: v.clone() for k, v in w.items()}
    y_ref = orc.resnet152_features(w_ref, x, train_bn=train, layers=layers)
    wd = _dev(w)
    runner = native.ResNetRunner(wd, layers, conv_mode=mode)
    y = runner.forward(x.to(DEV), train_bn=train)
    _close("features", y, y_ref, 5e-4)
    for k in w:
        if "running" in k:
            if train:
                _close(k, wd[k], w_ref[k], 2e-4, atol=1e-6)
            else:
                assert torch.equal(wd[k].cpu(), w[k])


@pytest.mark.parametrize("mode", ["fp32", "bf16x3", "f16x2"])
def test_resnet152_full_depth(lib, mode):
    """All 155 conv+BN layers at 224x224 (B=2), batch-statistics mode (quirk Q1).  fp32 rounding is amplified by
    152 layers of batch-statistics BatchNorm over a 2-image batch, so the yardstick is an fp64 evaluation of the
    oracle: the GPU result must sit inside the same error envelope as the fp32 CPU path (<= 4x its error)."""
    w = syn.resnet152_weights(seed=125)
    x = syn.rgb_images(2, seed=123)
    y_ref = orc.resnet152_features({k: v.clone() for k, v in w.items()}, x, train_bn=True)
    y64 = orc.resnet152_features({k: v.double() for k, v in w.items()}, x.double(), train_bn=True)
    runner = native.ResNetRunner(_dev(w), conv_mode=mode)
    y = runner.forward(x.to(DEV), train_bn=True)
    assert y.shape == (2, 196, 2048)
    y4 = y.reshape(2, 7, 2, 7, 2, 2048)
    assert torch.equal(y4[:, :, 0, :, 0], y4[:, :, 1, :, 1]), "7x7 -> 14x14 must be exact 2x2 replication (Q3)"
    scale = float(y64.abs().max())
    err_cpu32 = float((y_ref.double() - y64).abs().max()) / scale
    err_gpu = float((y.cpu().double() - y64).abs().max()) / scale
    print(f"ResNet-152 error vs fp64: CPU fp32 oracle {err_cpu32:.2e}, HIP[{mode}] {err_gpu:.2e}")
    assert err_gpu <= max(4.0 * err_cpu32, 5e-4), (err_gpu, err_cpu32)
    _close("features", y, y_ref, 5e-3)


@pytest.mark.parametrize("mode", ["bf16x3", "f16x2"])
def test_resnet_forward_with_bn_apply_folded_into_1x1_convs_matches_plane_route(lib, mode):
    """Batch-64 ResNet-152 forward (the size at which the launch policy puts the 1x1 convolutions on the persistent kernel), train-mode
    BatchNorm: with the BatchNorm-apply / residual / ReLU / split passes folded into the operand path of the consuming 1x1 convolutions
    (switch 103, the default; 101 / 102 = block inputs only / conv3 inputs only) the feature map and every running statistic must agree
    with the route that writes every convolution input as planes first (switch 100) bit for bit: the element-wise arithmetic (one
    fused multiply-add, one add, max, the round-to-nearest-even three-way split) and the order of the products are the same, only the
    kernel that performs them differs.  Both operand formats (f16x2: the pass and the producer waves scale by 4 and split into two
    fp16 planes with the same two roundings; an identity then always travels as fp32)."""
    w = syn.resnet152_weights(seed=125)
    imgs = syn.rgb_images(64, seed=123).to(DEV)
    results = {}
    try:
        # (96: layer 1's 64-channel conv1 stays on the plane route here - its plane kernel, the 64x64 tile form, and the persistent
        #  kernel that takes it under switch 97 order the three f16x2 products of a k-step differently, so that folding is compared
        #  at rounding level below, not bit for bit)
        assert lib.dic_debug_force_staged_gemm(96) == 0
        for code in (100, 101, 102, 103, 103):
            assert lib.dic_debug_force_staged_gemm(code) == 0
            wd = _dev(w)
            y = native.ResNetRunner(wd, conv_mode=mode).forward(imgs, train_bn=True, compact=True)
            torch.cuda.synchronize()
            assert torch.isfinite(y).all()
            stats = torch.cat([wd[k].flatten() for k in sorted(wd) if "running" in k])
            if code in results:
                assert torch.equal(results[code][0], y) and torch.equal(results[code][1], stats), "switch 103 does not reproduce itself"
            results[code] = (y.clone(), stats.clone())
        if mode == "f16x2":       # the default folding with layer 1's conv1 on the on-the-fly kernel as well (97)
            assert lib.dic_debug_force_staged_gemm(104) == 0 and lib.dic_debug_force_staged_gemm(97) == 0
            wd = _dev(w)
            y = native.ResNetRunner(wd, conv_mode=mode).forward(imgs, train_bn=True, compact=True)
            torch.cuda.synchronize()
            results[97] = (y.clone(), torch.cat([wd[k].flatten() for k in sorted(wd) if "running" in k]).clone())
    finally:
        lib.dic_debug_force_staged_gemm(104)
        lib.dic_debug_force_staged_gemm(97)
    y0, s0 = results[100]
    scale = float(y0.abs().max())
    if 97 in results:
        dy, ds = float((results[97][0] - y0).abs().max()) / scale, float((results[97][1] - s0).abs().max()) / float(s0.abs().max())
        print(f"default folding incl. layer 1 (97) vs 100: features max |d| / max = {dy:.2e}, running statistics {ds:.2e}")
        assert dy < 2e-3 and ds < 1e-4, (97, dy, ds)
    for code in (101, 102, 103):
        y, st = results[code]
        dy, ds = float((y - y0).abs().max()) / scale, float((st - s0).abs().max()) / float(s0.abs().max())
        print(f"switch {code} vs 100: features max |d| / max = {dy:.2e}, running statistics {ds:.2e}")
        if mode == "f16x2" and code in (102, 103):
            # conv3 on the on-the-fly kernel runs with the remainder-round K split, on the plane route (f16x2) on the twelve-wave
            # kernel without it: same products, another association of the partial sums - rounding level, amplified by the
            # BatchNorm chain like any reordering (the f16x2 default folds block inputs only: 101 == 104 is the exact statement)
            assert dy < 2e-3 and ds < 1e-4, (code, dy, ds)
        else:
            assert torch.equal(y, y0) and torch.equal(st, s0), (code, dy, ds)


@pytest.mark.parametrize("B,size", [(16, 224), (3, 130), (2, 520)])
def test_depth_encoder_layer1_sparse_backward_equals_dense(lib, B, size):
    """Round 4: the backward of the depth encoder's first layer without its full-size gradient (csrc/depth_layer1.hip, switch 181,
    default) against the three dense passes it replaces (180) on the same tape: the gradients of conv1.weight, bn1.weight and bn1.bias
    agree to 2e-5 of each tensor's scale (both evaluate the same sums; the sparse form keeps the long ones in fp64), every other
    gradient is bit-identical (nothing upstream of layer 1 changed), and conv1.bias - whose true gradient in front of a train-mode
    BatchNorm is exactly zero - is exactly zero instead of rounding noise.  Bench shape, a small odd map and a wide one."""
    enc, st = syn.depth_encoder_weights(seed=124)
    depth = syn.depth_maps(B, seed=123, size=size).to(DEV)
    f, tape = native.depth_encoder_forward(_dev(enc), _dev(st), depth, train=True, compact=(size == 224))
    g = torch.Generator().manual_seed(5)
    dfeat = (torch.randn(f.shape, generator=g) * 1e-2).to(DEV)
    out = {}
    try:
        for code in (180, 181):
            assert lib.dic_debug_force_staged_gemm(code) == 0
            grads = native.depth_encoder_backward(tape, dfeat)
            torch.cuda.synchronize()
            out[code] = {k: v.clone() for k, v in grads.items()}
    finally:
        lib.dic_debug_force_staged_gemm(181)
    for k in out[180]:
        a, b = out[180][k], out[181][k]
        assert torch.isfinite(b).all(), k
        if k == "conv1.bias":
            assert not bool(b.any()), "the sparse form writes the exactly-zero bias gradient as zero"
        elif k in ("conv1.weight", "bn1.weight", "bn1.bias"):
            d, sc = float((a - b).abs().max()), float(a.abs().max())
            print(f"{k}: max |sparse - dense| / scale = {d / sc:.2e}")
            assert d <= 2e-5 * sc, (k, d, sc)
        else:
            assert torch.equal(a, b), k


def test_resnet_forward_with_bn_apply_inside_the_halo_kernel_matches_plane_route(lib):
    """Round 4: in the f16x2 format the 3x3 convolutions of 14x14 maps (35 of the 50 blocks) read the RAW output of conv1 and form
    relu(bn1(.)) in the LDS-halo kernel's producer waves (switch 109, default) instead of reading planes written by a bn_apply_planes
    pass (108).  Same element-wise arithmetic (one fused multiply-add, max, scale by 4, the two fp16 roundings), same LDS image, same
    products in the same order: features and every running statistic agree bit for bit, the guard word stays clear, and the
    default reproduces itself.  The 28x28 maps of layer 2 (switch 95, default) take the same kernel in its 9-row geometry instead of
    the gathered 3x3 kernel + planes pass (94): compared at rounding level."""
    w = syn.resnet152_weights(seed=125)
    imgs = syn.rgb_images(64, seed=123).to(DEV)
    results = {}
    try:
        for code, code28 in ((108, 94), (109, 94), (109, 95), (109, 95)):
            assert lib.dic_debug_force_staged_gemm(code) == 0 and lib.dic_debug_force_staged_gemm(code28) == 0
            code = (code, code28)
            wd = _dev(w)
            runner = native.ResNetRunner(wd, conv_mode="f16x2")
            y = runner.forward(imgs, train_bn=True, compact=True)
            torch.cuda.synchronize()
            assert torch.isfinite(y).all() and int(runner.status_word().item()) == 0
            stats = torch.cat([wd[k].flatten() for k in sorted(wd) if "running" in k])
            if code in results:
                assert torch.equal(results[code][0], y) and torch.equal(results[code][1], stats), "the default does not reproduce itself"
            results[code] = (y.clone(), stats.clone())
    finally:
        lib.dic_debug_force_staged_gemm(109)
        lib.dic_debug_force_staged_gemm(95)
    assert torch.equal(results[(108, 94)][0], results[(109, 94)][0]) and torch.equal(results[(108, 94)][1], results[(109, 94)][1])
    # layer 2 (28x28 maps, switch 95): the halo kernel sums chunk-major where the gathered kernel it replaces sums tap-major - same
    # products, another association, amplified by the BatchNorm chain like any reordering (cf. the test above)
    y0, s0 = results[(109, 94)]
    y1, s1 = results[(109, 95)]
    dy, ds = float((y1 - y0).abs().max()) / float(y0.abs().max()), float((s1 - s0).abs().max()) / float(s0.abs().max())
    print(f"layer-2 halo route vs gathered route: features max |d| / max = {dy:.2e}, running statistics {ds:.2e}")
    assert dy < 2e-3 and ds < 1e-4, (dy, ds)


def test_resnet_forward_downsample_bn_inside_the_1x1_kernel_is_bit_identical(lib):
    """Round 4 (f16x2): the residual of a stage's second block is the downsample branch's RAW output; its BatchNorm is one fused
    multiply-add per element, done by the producer waves of the on-the-fly 1x1 kernel (switch 99, default) instead of an in-place
    pass over the branch (98).  Same value, same rounding: features and running statistics agree bit for bit."""
    w = syn.resnet152_weights(seed=125)
    imgs = syn.rgb_images(64, seed=123).to(DEV)
    out = {}
    try:
        for code in (98, 99):
            assert lib.dic_debug_force_staged_gemm(code) == 0
            wd = _dev(w)
            runner = native.ResNetRunner(wd, conv_mode="f16x2")
            y = runner.forward(imgs, train_bn=True, compact=True)
            torch.cuda.synchronize()
            assert torch.isfinite(y).all() and int(runner.status_word().item()) == 0
            out[code] = (y.clone(), torch.cat([wd[k].flatten() for k in sorted(wd) if "running" in k]).clone())
    finally:
        lib.dic_debug_force_staged_gemm(99)
    assert torch.equal(out[98][0], out[99][0]) and torch.equal(out[98][1], out[99][1])


def test_resnet_forward_with_interleaved_fragment_reads_is_bit_identical(lib):
    """Round 4 (f16x2): the computing waves of the 128x128 kernels (LDS-halo 3x3, on-the-fly 1x1, gathered, row-major) issue one
    fragment read in the gap behind each matrix instruction (switch 121, default) instead of a block of reads in front of each k-step's
    matrix instructions (120).  Same products in the same order per accumulator: features and running statistics agree bit for bit."""
    w = syn.resnet152_weights(seed=126)
    imgs = syn.rgb_images(64, seed=124).to(DEV)
    out = {}
    try:
        for code in (120, 121):
            assert lib.dic_debug_force_staged_gemm(code) == 0
            wd = _dev(w)
            runner = native.ResNetRunner(wd, conv_mode="f16x2")
            y = runner.forward(imgs, train_bn=True, compact=True)
            torch.cuda.synchronize()
            assert torch.isfinite(y).all() and int(runner.status_word().item()) == 0
            out[code] = (y.clone(), torch.cat([wd[k].flatten() for k in sorted(wd) if "running" in k]).clone())
    finally:
        lib.dic_debug_force_staged_gemm(121)
    assert torch.equal(out[120][0], out[121][0]) and torch.equal(out[120][1], out[121][1])


def test_resnet_forward_single_launch_batchnorm_finalize_for_layer2_is_bit_identical(lib):
    """Round 4: train-mode BatchNorm statistics of a layer with 513..1024 rows of partial sums (ResNet layer 2 at batch 64: 784) are
    finalized by ONE launch (switch 183, default) instead of slice sums + finalize (182).  Both add the same fp32 partials in fp64 - where
    sums of a few hundred fp32 values are exact whatever the order - so features and running statistics agree bit for bit."""
    w = syn.resnet152_weights(seed=127)
    imgs = syn.rgb_images(64, seed=125).to(DEV)
    out = {}
    try:
        for code in (182, 183):
            assert lib.dic_debug_force_staged_gemm(code) == 0
            wd = _dev(w)
            runner = native.ResNetRunner(wd, conv_mode="f16x2")
            y = runner.forward(imgs, train_bn=True, compact=True)
            torch.cuda.synchronize()
            assert torch.isfinite(y).all() and int(runner.status_word().item()) == 0
            out[code] = (y.clone(), torch.cat([wd[k].flatten() for k in sorted(wd) if "running" in k]).clone())
    finally:
        lib.dic_debug_force_staged_gemm(183)
    assert torch.equal(out[182][0], out[183][0]) and torch.equal(out[182][1], out[183][1])


def test_layer1_kernels_reproducible_next_to_lds_heavy_kernels(lib):
    """The packed-FMA layer-1 kernels of the depth encoder (csrc/conv1_depth.hip), called alone through the library, repeated
    on identical inputs while a bf16x3 ResNet forward on its round-1 gather kernels (debug codes 70 75: three LDS-heavy
    workgroups per CU) runs on a second stream.  Their first form differed in 59 of 59 such repetitions (one pixel in 1e5,
    channels 48..63); see the note at the top of the file for what was wrong.  Every repetition must be bit-identical."""
    import ctypes as C
    import time
    from depth_image_captioning_pub_amd._lib import ptr
    B = 64
    x = syn.depth_maps(B, seed=123).to(DEV)
    g = torch.Generator().manual_seed(1)
    dy = (torch.randn(B, 73, 73, 128, generator=g) * 1e-3).to(DEV)
    w = (torch.randn(128, 49, generator=g) * 0.1).to(DEV)
    bias = torch.zeros(128, device=DEV)
    ws = torch.zeros(1024 * 6400, device=DEV); cs = torch.zeros(256 * 2048 * 4, device=DEV)
    dw = torch.zeros(128 * 49, device=DEV); db = torch.zeros(128, device=DEV)
    y = torch.zeros(B, 73, 73, 128, device=DEV); part = torch.zeros(1100 * 2 * 128, device=DEV)
    layers = (1, 1, 1, 1)
    hog_stream = torch.cuda.Stream()
    try:
        for code in (70, 75):
            lib.dic_debug_force_staged_gemm(code)
        rn = native.ResNetRunner({k: v.to(DEV) for k, v in syn.resnet152_weights(seed=125, layers=layers).items()}, layers,
                                 conv_mode="bf16x3")
        imgs = syn.rgb_images(64, seed=123).to(DEV)
        with torch.cuda.stream(hog_stream):
            rn.forward(imgs, True, compact=True)
        torch.cuda.synchronize()
        main = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        ref = None
        for rep in range(30):
            with torch.cuda.stream(hog_stream):
                for _ in range(6):
                    rn.forward(imgs, True, compact=True)
            time.sleep(0.004)                  # let the side stream get going before the kernels under test are enqueued
            assert lib.dic_debug_conv1_fwd(ptr(x), B, 224, 224, ptr(w), ptr(bias), ptr(y), ptr(part), main) == 0
            assert lib.dic_debug_conv1_wgrad(ptr(x), B, 224, 224, ptr(dy), ptr(dw), ptr(db), ptr(ws), ptr(cs), main) == 0
            torch.cuda.synchronize()
            cur = (y.clone(), part[:512 * 256].clone(), dw.clone(), db.clone())
            if ref is None:
                ref = cur
                assert torch.isfinite(cur[0]).all() and torch.isfinite(cur[2]).all()
                continue
            for name, a, b in zip(("y", "BatchNorm partials", "dW", "db"), cur, ref):
                assert torch.equal(a, b), f"repetition {rep}: {name} differs from the first repetition"
    finally:
        lib.dic_debug_force_staged_gemm(79)
        lib.dic_debug_force_staged_gemm(78)
        torch.cuda.synchronize()


def test_depth_encoder_f16x2_equals_bf16x3_at_fp32_level(lib):
    """Round 4: conv2 / conv3 of the depth encoder (forward, data gradient, weight gradient) run in the f16x2 operand format with scales
    chosen on the device (weights: exact maximum; gradients: the bound their BatchNorm backward computes before splitting), switch 117
    (default), against the exact bf16x3 split of rounds 1-3 (116) on the bench shape: features agree to 2e-5 of their scale; the 12
    gradients to 2e-3 in relative L2 norm - no per-element bound, because the two forwards may break max-pool /
    ReLU ties differently and a flipped tie re-routes one gradient element (measured here: isolated differences of 7e-4 of scale in
    grad.conv1.weight, 3e-3 in grad.bn1.bias and 2.5e-2 in one row of grad.conv3.weight - one term of a 784-term sum - at relative L2
    distances of at most 7e-4; the direct oracle comparisons above see the same).  The tight
    statement is the one of the *_decision_replay tests above, which run in the default format: with the selections replayed, 2e-4
    against fp64.  Within one format the selections are the same for every upstream scale, so linearity is checked to 1e-4,
    tiny gradients included: the upstream gradient is scaled by 1e-6 and by 1e+4 to show that no range assumption is made."""
    B = 16
    enc, st = syn.depth_encoder_weights(seed=124)
    depth = syn.depth_maps(B, seed=123).to(DEV)
    g = torch.Generator().manual_seed(4)
    dfeat0 = torch.randn(B, 49, 2048, generator=g).to(DEV)
    out = {}
    try:
        for code in (116, 117):
            assert lib.dic_debug_force_staged_gemm(code) == 0
            for mag in (1.0, 1e-6, 1e4):
                f, tape = native.depth_encoder_forward(_dev(enc), _dev(st), depth, train=True, compact=True)
                grads = native.depth_encoder_backward(tape, dfeat0 * mag)
                torch.cuda.synchronize()
                assert torch.isfinite(f).all() and int(native.depth_status_word(tape).item()) == 0
                out[(code, mag)] = (f.clone(), {k: v.clone() for k, v in grads.items()})
    finally:
        lib.dic_debug_force_staged_gemm(117)
    for mag in (1.0, 1e-6, 1e4):
        f3, g3 = out[(116, mag)]
        f2, g2 = out[(117, mag)]
        assert float((f2 - f3).abs().max()) <= 2e-5 * float(f3.abs().max())
        for k in g3:
            if k.startswith("conv") and k.endswith("bias"):
                continue                              # exactly-zero true gradient in front of train-mode BatchNorm (Q10): noise on both sides
            d, sc = float((g2[k] - g3[k]).abs().max()), float(g3[k].abs().max())
            l2 = float((g2[k] - g3[k]).norm() / g3[k].norm())
            assert torch.isfinite(g2[k]).all() and l2 <= 2e-3, (mag, k, d, sc, l2)
        # linearity in the upstream gradient survives the per-step scale choice
        if mag != 1.0:
            for k in ("conv2.weight", "conv3.weight", "bn1.weight"):
                ref = out[(117, 1.0)][1][k] * mag
                assert float((g2[k] - ref).abs().max()) <= 1e-4 * float(ref.abs().max()), (mag, k)
