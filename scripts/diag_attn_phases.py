"""Phase time stamps of attn_fwd_kernel<49> (s_memtime; experiments library): where do the ~14 us of a decode step's attention
launch go?  Runs the decoder forward at bench shape (batch 64, T 20, V 10000, compact 49 cells) and prints, for workgroup 0 and
a middle workgroup of the LAST step's launch, the time between consecutive stamps in us (s_memtime ticks at 100 MHz)."""
import os, sys, ctypes as C
os.environ.setdefault("DIC_LIB", "experiments")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_image_captioning_pub_amd import native, synthetic as syn, _lib
lib = _lib.load(); DEV = "cuda:0"
B, V, T = 64, 10000, 20
w = {k: v.to(DEV) for k, v in syn.decoder_weights(V, seed=123).items()}
f = syn.features(B, 5)[:, :49].contiguous().to(DEV); fd = syn.features(B, 6)[:, :49].contiguous().to(DEV)
caps, lens = syn.captions_fixed(B, V, T, seed=123); caps = caps.to(DEV)
drop = syn.dropout_multiplier(B, T, 0.5, seed=1).to(DEV)
names = ["start->LSTM+barrier", "q partial+barrier", "q final+barrier", "scores (+F/gate loads issued)", "barrier", "softmax", "ctx+gate", "barrier", "store"]
ws = None
for rep in range(3):
    logits, alphas, tape = native.decoder_forward(w, f, fd, caps, lens, drop, workspace=ws)
    ws = tape.workspace
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 32)()
    _lib.check(lib.dic_debug_attn_stamps(buf))
    for g in range(2):
        st = [buf[g * 16 + i] for i in range(10)]
        d = [(st[i + 1] - st[i]) / 100.0 for i in range(9)]
        print(f"rep {rep} workgroup {'0' if g == 0 else '257'}: total {(st[9] - st[0]) / 100.0:6.2f} us  " + "  ".join(f"{n}: {x:.2f}" for n, x in zip(names, d)))
