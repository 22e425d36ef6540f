"""Decoder stage timing (forward / loss / backward), persistent forward loop vs per-step launches, both layouts.
usage: python scripts/bench_decoder.py [B] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_image_captioning_pub_amd import native, synthetic as syn, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
V, T, dev = 10000, 20, "cuda:0"
lib = _lib.load()
w = {k: v.to(dev) for k, v in syn.decoder_weights(V, seed=123).items()}
caps, lens = syn.captions_fixed(B, V, T, seed=123)
caps = caps.to(dev)
drop = syn.dropout_multiplier(B, T, 0.5, seed=123).to(dev)
grads = {k: torch.empty_like(v) for k, v in w.items()}
for cells in (49, 196):
    f = syn.features(B, 5)
    if cells == 49:
        f = f.reshape(B, 14, 14, 2048)[:, ::2, ::2].reshape(B, 49, 2048).contiguous()
    fr, fd = f.to(dev), (0.5 * f).to(dev)
    ref = None
    for mode in (140, 141):
        lib.dic_debug_force_staged_gemm(mode)
        ws = None
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        tf = tl = tb = 0.0
        for it in range(iters + 3):
            ev[0].record()
            logits, alphas, tape = native.decoder_forward(w, fr, fd, caps, lens, drop, workspace=ws)
            ws = tape.workspace
            ev[1].record()
            loss, dl, da = native.caption_loss(logits, native.pack_targets(caps, lens), alphas, in_place=True)
            ev[2].record()
            native.decoder_backward(tape, dl, da, grads=grads)
            ev[3].record()
            torch.cuda.synchronize()
            if it >= 3:
                tf += ev[0].elapsed_time(ev[1]); tl += ev[1].elapsed_time(ev[2]); tb += ev[2].elapsed_time(ev[3])
        chk = (float(loss.item()), float(grads["decode_step.weight_ih"].double().abs().sum()))
        if ref is None:
            ref = chk
        print(f"cells {cells:3d} {'persistent' if mode == 141 else 'per-step  '}: fwd {tf / iters:.3f} ms  loss {tl / iters:.3f}  bwd {tb / iters:.3f}  "
              f"total {(tf + tl + tb) / iters:.3f} | loss {chk[0]:.6f} (per-step {ref[0]:.6f}) grad-abs {chk[1]:.6e} ({ref[1]:.6e})", flush=True)
lib.dic_debug_force_staged_gemm(141)
