"""GPU: the overflow guard of the f16x2 operand format (csrc/common.h, include/dic.h "OVERFLOW GUARD").

The format stores 4 * x in fp16 planes, so a layer input beyond +-16376 becomes inf in the first plane and -inf in the second;
their products sum to NaN, and every ReLU downstream is fmaxf(v, 0), which turns that NaN into 0 - the VALUES of the output can
look perfectly sane.  What keeps the failure loud is a status word raised by every kernel that writes planes:
  * dic_split_f16x2_paired_checked raises the caller's word (the DPT front-end checks it once per forward);
  * dic_resnet_fwd clears / raises the first word of its workspace and fills the features with NaN when it is raised;
  * engine.CaptionTrainer hands the word to AdamW and to the BatchNorm running-statistic update (both skip on the device) and
    raises DicError when the host next looks (check_status / the following steps);
  * the CNNEncoder_Atten shim raises after the forward.
The reference itself has no such failure mode (plain fp32, Base_caption_model/base_caption_models.py:36-45): the guard exists so that
the faster arithmetic can be the default without ever returning numbers the reference would not have produced."""
import ctypes as C

import pytest
import torch

from depth_image_captioning_pub_amd import _lib, native, synthetic as syn
from depth_image_captioning_pub_amd._lib import DicError, check, ptr, stream_ptr
from depth_image_captioning_pub_amd.engine import CaptionTrainer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TINY = (1, 1, 1, 1)
HOT = "backbone.4.0.bn1."      # BatchNorm behind the first bottleneck's conv1: gamma = 1e5 puts its output at ~1e5 * z-score


def _split_checked(lib, x, scale):
    rows, k = x.shape
    n = (rows + 1) // 2 * 2 * k
    h1, h2 = (torch.empty(n, dtype=torch.int16, device=DEV) for _ in range(2))
    word = torch.zeros(1, dtype=torch.int32, device=DEV)
    check(lib.dic_split_f16x2_paired_checked(ptr(x), C.c_longlong(rows), k, C.c_float(scale), ptr(h1), ptr(h2), ptr(word), stream_ptr()),
          "dic_split_f16x2_paired_checked")
    return h1, h2, int(word.item())


def test_split_raises_the_word_exactly_when_a_value_leaves_the_fp16_range(lib):
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(64, 96, generator=g) * 16000.0).to(DEV)            # 4 * x < 64000: fits
    assert _split_checked(lib, x, 4.0)[2] == 0
    x[17, 40] = 16376.0                                                # the documented bound itself: 65504, the largest finite fp16
    assert _split_checked(lib, x, 4.0)[2] == 0
    for bad in (16380.0, -2.0e4, float("inf"), float("nan")):
        y = x.clone()
        y[17, 40] = bad
        assert _split_checked(lib, y, 4.0)[2] != 0, bad


def test_overflowed_activation_is_hidden_by_relu_but_not_from_the_guard(lib):
    """feed an activation above 16376 through dic_conv2d_f16x2 followed by ReLU.  The convolution's output is NaN where the
    bad pixel enters, the fused ReLU (fmaxf) makes it 0: a finite, plausible map - only the split's guard word knows."""
    B, H, W, Cin, CO = 1, 8, 8, 32, 128
    g = torch.Generator().manual_seed(9)
    x = torch.rand(B * H * W, Cin, generator=g).to(DEV)
    x[20, 3] = 2.0e4
    w = (torch.randn(CO, Cin, generator=g) * 0.1).to(DEV)
    wmax = float(w.abs().max())
    import math
    ws = 2.0 ** math.floor(14 - math.log2(wmax))
    xh1, xh2, raised = _split_checked(lib, x, 4.0)
    wh1, wh2, w_raised = _split_checked(lib, w, ws)
    assert raised != 0 and w_raised == 0
    y = torch.full((B * H * W, CO), 7.0, device=DEV)
    xp = (C.c_void_p * 2)(xh1.data_ptr(), xh2.data_ptr())
    wp = (C.c_void_p * 2)(wh1.data_ptr(), wh2.data_ptr())
    check(lib.dic_conv2d_f16x2(xp, B, H, W, Cin, wp, None, CO, 1, 1, 1, 0, 1, ptr(y), None, C.c_float(1.0 / (4.0 * ws)), stream_ptr()),
          "dic_conv2d_f16x2")
    torch.cuda.synchronize()
    assert torch.isfinite(y).all() and bool((y[20] == 0).all()), "the ReLU no longer hides the overflow: revisit this test's premise"


def _hot_resnet(mode):
    w = {k: v.to(DEV) for k, v in syn.resnet152_weights(seed=41, layers=TINY).items()}
    w[HOT + "weight"].fill_(1.0e5)
    return w, native.ResNetRunner(w, TINY, conv_mode=mode)


def test_resnet_forward_raises_the_word_and_poisons_the_features(lib):
    imgs = syn.rgb_images(4, seed=3, size=96).to(DEV)
    w, rn = _hot_resnet("f16x2")
    before = {k: v.clone() for k, v in w.items() if "running" in k}
    y = rn.forward(imgs, train_bn=True)
    torch.cuda.synchronize()
    assert int(rn.status_word().item()) != 0
    assert bool(torch.isnan(y).all()), "features of a flagged forward must be NaN everywhere"
    with pytest.raises(DicError, match="f16x2"):
        rn.check_overflow()
    for k, v in before.items():                      # no running statistic may have turned non-finite
        assert torch.isfinite(w[k]).all(), k
    # the layers in front of the overflow were updated as usual, the ones fed by it were left alone
    assert not torch.equal(w["backbone.1.running_mean"], before["backbone.1.running_mean"])
    # ... the same weights and images in the exact-operand format are fine (the guard is about the format, not the network)
    _, rn3 = _hot_resnet("bf16x3")
    y3 = rn3.forward(imgs, train_bn=True)
    assert torch.isfinite(y3).all() and int(rn3.status_word().item()) == 0
    rn3.check_overflow()
    # ... and a healthy f16x2 forward clears the word again
    w[HOT + "weight"].fill_(1.0)
    rn_ok = native.ResNetRunner(w, TINY, conv_mode="f16x2")
    y_ok = rn_ok.forward(imgs, train_bn=True)
    assert torch.isfinite(y_ok).all() and int(rn_ok.status_word().item()) == 0


@pytest.mark.parametrize("prefetch", [False, True])
def test_trainer_skips_the_update_on_the_device_and_raises(lib, prefetch):
    """inject an overflow into a ResNet activation; the step must raise and leave weights, Adam moments and
    the BatchNorm running statistics untouched - without a host synchronisation inside train_step."""
    vocab, B = 60, 4
    rn = syn.resnet152_weights(seed=41, layers=TINY)
    tr = CaptionTrainer(vocab, device=DEV, resnet_layers=TINY, seed=2, resnet_init=rn)      # default arithmetic
    assert tr.conv_mode == native.DEFAULT_CONV_MODE == "f16x2"
    imgs = [syn.rgb_images(B, seed=70 + i, size=96).to(DEV) for i in range(3)]
    depth = syn.depth_maps(B, seed=70, size=96).to(DEV)
    caps, lens = syn.captions_fixed(B, vocab, 6, seed=70)
    caps = caps.to(DEV)
    nxt = (lambda i: {"next_imgs": imgs[i]}) if prefetch else (lambda i: {})
    l0 = tr.train_step(imgs[0], depth, caps, lens, **nxt(1))                  # a healthy step first
    tr.check_status()
    assert torch.isfinite(l0).all() and tr.step_count == 1
    if prefetch:                                                             # the forward of imgs[1] is already in flight: let it finish healthy
        torch.cuda.synchronize()
    snap = {k: getattr(tr.flat, k).clone() for k in ("data", "exp_avg", "exp_avg_sq")}
    stats = tr.rn_stats.clone()
    tr.rn_w[HOT + "weight"].fill_(1.0e5)                                      # from here on every forward overflows behind this layer
    if prefetch:
        l1 = tr.train_step(imgs[1], depth, caps, lens, **nxt(2))              # healthy features (computed before the injection) ...
        l2 = tr.train_step(imgs[2], depth, caps, lens)                        # ... the prefetched forward of imgs[2] is the bad one
        snap_ok = None
    else:
        l2 = tr.train_step(imgs[1], depth, caps, lens)
    # nothing above synchronised; the guard tripped on the device
    with pytest.raises(DicError, match="overflow guard"):
        tr.check_status()
    assert bool(torch.isnan(l2).all()), "the loss of the flagged step must be NaN"
    if not prefetch:
        assert tr.step_count == 1, "Adam's step count must not include the skipped update"
        for k, v in snap.items():
            assert torch.equal(getattr(tr.flat, k), v), f"{k} changed although the update had to be skipped"
        # running statistics: layers in front of the injected one may advance (the reference would have advanced them too), no entry
        # may be non-finite, and the deferred / in-place update never wrote NaN
        assert torch.isfinite(tr.rn_stats).all()
    else:
        assert tr.step_count == 2 and torch.isfinite(l1).all()
        assert torch.isfinite(tr.flat.data).all() and torch.isfinite(tr.flat.exp_avg).all() and torch.isfinite(tr.flat.exp_avg_sq).all()
        assert torch.isfinite(tr.rn_stats).all()
    # the trainer stays usable once the cause is gone
    tr.rn_w[HOT + "weight"].fill_(1.0)
    tr.prefetched = None
    l3 = tr.train_step(imgs[0], depth, caps, lens)
    tr.check_status()
    assert torch.isfinite(l3).all()


def test_encoder_shim_raises_in_the_default_arithmetic(lib):
    from depth_image_captioning_pub_amd.Captioning_models.Base_caption_model.base_caption_models import CNNEncoder_Atten
    enc = CNNEncoder_Atten(14, layers=TINY)
    assert enc.conv_mode == native.DEFAULT_CONV_MODE
    sd = enc.state_dict()
    sd.update(syn.resnet152_weights(seed=41, layers=TINY))
    enc.load_state_dict(sd)
    enc.to(DEV).train()
    x = syn.rgb_images(2, seed=5, size=96).to(DEV)
    assert torch.isfinite(enc(x)).all()
    with torch.no_grad():
        enc.backbone[4][0].bn1.weight.fill_(1.0e5)
    with pytest.raises(DicError, match="f16x2"):
        enc(x)
    enc.conv_mode = "bf16x3"                                 # the documented remedy
    assert torch.isfinite(enc(x)).all()


def test_dpt_runner_guard_word(lib):
    """... and through DptRunner(arith='f16x2').  A huge (non-standardised) convolution weight in the decoder puts the
    next layer's input beyond the fp16 range; the final ReLUs would return a finite map."""
    from depth_image_captioning_pub_amd import dpt
    cfg = syn.DptConfig(layers=(1, 1, 1), depth=2, hooks=(0, 1))
    w = {k: v.to(DEV) for k, v in syn.dpt_weights(8, cfg).items()}
    x = syn.dpt_images(1, seed=3, size=64).to(DEV)
    ok = dpt.DptRunner(w, cfg, arith="f16x2")
    assert torch.isfinite(ok.forward(x)).all() and int(ok.overflow.item()) == 0
    w2 = dict(w)
    w2["scratch.layer1_rn.weight"] = w["scratch.layer1_rn.weight"] * 3.0e6
    bad = dpt.DptRunner(w2, cfg, arith="f16x2")
    with pytest.raises(DicError, match="fp16 range"):
        bad.forward(x)
    assert int(bad.overflow.item()) != 0
    assert torch.isfinite(dpt.DptRunner(w2, cfg, arith="bf16x3").forward(x)).all()


def test_depth_encoder_guard_word(lib):
    """The depth encoder's conv2 / conv3 read f16x2 planes of 4 * (pooled BatchNorm output): a BatchNorm gain that pushes it beyond
    16376 raises the status word at offset 0 of its workspace and turns the features into NaN; the engine ORs that word into the
    step's guard (AdamW skipped on the device, DicError from check_status)."""
    enc, st = syn.depth_encoder_weights(seed=124)
    enc = {k: v.to(DEV) for k, v in enc.items()}
    st = {k: v.to(DEV) for k, v in st.items()}
    depth = syn.depth_maps(4, seed=9).to(DEV)
    f, tape = native.depth_encoder_forward(enc, st, depth, train=True, compact=True)
    assert torch.isfinite(f).all() and int(native.depth_status_word(tape).item()) == 0
    enc["bn1.weight"].fill_(1.0e5)
    f, tape = native.depth_encoder_forward(enc, st, depth, train=True, compact=True)
    torch.cuda.synchronize()
    assert int(native.depth_status_word(tape).item()) != 0 and bool(torch.isnan(f).all())
    # through the engine
    vocab = 60
    tr = CaptionTrainer(vocab, device=DEV, resnet_layers=TINY, seed=2)
    imgs = syn.rgb_images(4, seed=70, size=224).to(DEV)
    caps, lens = syn.captions_fixed(4, vocab, 6, seed=70)
    tr.train_step(imgs, depth, caps.to(DEV), lens)
    tr.check_status()
    snap = tr.flat.data.clone()
    tr.enc_w["bn1.weight"].fill_(1.0e5)
    snap = tr.flat.data.clone()
    tr.train_step(imgs, depth, caps.to(DEV), lens)
    with pytest.raises(DicError, match="overflow guard"):
        tr.check_status()
    assert torch.equal(tr.flat.data, snap), "parameters changed although the update had to be skipped"
