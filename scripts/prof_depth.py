"""rocprofv3 target: depth-encoder forward + backward (batch 64, train mode, compact map) x10."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_image_captioning_pub_amd import native, synthetic as syn
DEV = "cuda:0"
from depth_image_captioning_pub_amd import _lib
for c in sys.argv[1:]: _lib.load().dic_debug_force_staged_gemm(int(c))
w, state = syn.depth_encoder_weights(seed=124)
w = {k: v.to(DEV) for k, v in w.items()}; state = {k: v.to(DEV) for k, v in state.items()}
depth = syn.depth_maps(64, seed=123).to(DEV)
ws = None
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
tf = tb = 0.0
for it in range(13):
    e0.record()
    feats, tape = native.depth_encoder_forward(w, state, depth, train=True, workspace=ws, compact=True)
    ws = tape.workspace
    e1.record()
    dfe = torch.ones_like(feats)
    grads = native.depth_encoder_backward(tape, dfe)
    e2.record(); torch.cuda.synchronize()
    if it >= 3: tf += e0.elapsed_time(e1); tb += e1.elapsed_time(e2)
print(f"depth encoder fwd {tf/10:.3f} ms, bwd {tb/10:.3f} ms")
