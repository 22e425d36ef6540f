"""Phase time stamps of the persistent decoder forward loop (workgroup g0 c0), microseconds per phase, averaged over steps."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_image_captioning_pub_amd import native, synthetic as syn, _lib
B, V, T, dev = 64, 10000, 20, "cuda:0"
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 49
lib = _lib.load()
lib.dic_debug_force_staged_gemm(141)
lib.dic_debug_force_staged_gemm(142 + (int(sys.argv[2]) if len(sys.argv) > 2 else 0))
w = {k: v.to(dev) for k, v in syn.decoder_weights(V, seed=123).items()}
caps, lens = syn.captions_fixed(B, V, T, seed=123); caps = caps.to(dev)
f = syn.features(B, 5)
if cells == 49:
    f = f.reshape(B, 14, 14, 2048)[:, ::2, ::2].reshape(B, 49, 2048).contiguous()
fr, fd = f.to(dev), (0.5 * f).to(dev)
buf = torch.zeros(T * 8, dtype=torch.int64, device=dev)
lib.dic_debug_decoder_stamps(C.c_void_p(buf.data_ptr()))
ws = None
for _ in range(5):
    logits, alphas, tape = native.decoder_forward(w, fr, fd, caps, lens, None, workspace=ws); ws = tape.workspace
torch.cuda.synchronize()
s = buf.cpu().view(T, 8).double() / 100.0          # us
names = ["wait", "lstm(slab read)", "-", "q", "scores(P)", "softmax+gate partial", "ctx", "partial gates(Wcat)+publish"]
d = s[1:, 1:] - s[1:, :-1]
print("per-phase us (mean over steps 1..T-1):")
for i in range(7):
    print(f"  {names[i]:32s} {float(d[:, i].mean()):6.2f}")
print("step period us:", float((s[2:, 0] - s[1:-1, 0]).mean()), " total loop us:", float(s[-1, 7] - s[0, 0]))
lib.dic_debug_decoder_stamps(C.c_void_p(0))
