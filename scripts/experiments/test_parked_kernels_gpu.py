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
