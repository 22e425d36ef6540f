"""Seed sweep behind the depth-encoder test sizes (tests/test_encoders_gpu.py): worst relative gradient error vs the oracle
for the generic MFMA layer-1 path (debug code 130) and the packed-FMA layer-1 kernels (132).  Small maps leave 18..288
samples per channel in layer 3, so one ReLU / max-pool decision that flips under 1e-6 rounding differences moves a
gradient by percents - for EITHER kernel; the test uses seeds on which no decision sits that close to a tie."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from depth_image_captioning_pub_amd import native, synthetic as syn, _lib
from oracle import captioning_oracle as orc
DEV="cuda:0"
lib=_lib.load()
def setup(size,B,seed):
    w, st = syn.depth_encoder_weights(seed=seed)
    g = torch.Generator().manual_seed(seed)
    for i in (1, 2, 3):
        w[f"bn{i}.weight"] = 1.0 + 0.2 * torch.randn(w[f"bn{i}.weight"].shape, generator=g)
        w[f"bn{i}.bias"] = 0.1 * torch.randn(w[f"bn{i}.bias"].shape, generator=g)
    depth = syn.depth_maps(B, seed=seed, size=size)
    d_out = torch.from_numpy(np.random.Generator(np.random.PCG64(seed + 7)).standard_normal((B, 196, 2048)).astype(np.float32)) * 1e-2
    return w, st, depth, d_out
def gpu(w, st, depth, d_out, code):
    lib.dic_debug_force_staged_gemm(code)
    y, tape = native.depth_encoder_forward({k:v.to(DEV) for k,v in w.items()}, {k:v.to(DEV) for k,v in st.items()}, depth.to(DEV), train=True)
    grads = native.depth_encoder_backward(tape, d_out.to(DEV))
    return {k:v.cpu() for k,v in grads.items()}
for size,B in ((109,2),(300,1),(520,1),(100,3)):
    for seed in range(52, 64):
        w, st, depth, d_out = setup(size,B,seed)
        st_ref = {k: v.clone() for k, v in st.items()}
        wg = {k: v.clone().requires_grad_(True) for k, v in w.items()}
        y_ref = orc.depth_encoder_forward(wg, st_ref, depth, train=True)
        (y_ref * d_out).sum().backward()
        res=[]
        for code in (130,132):
            g = gpu(w, st, depth, d_out, code)
            worst = max(float((g[k]-wg[k].grad).abs().max()/wg[k].grad.abs().max()) for k in w if not (k.startswith("conv") and k.endswith("bias")))
            res.append(worst)
        print(f"size {size} seed {seed}: worst rel grad err old {res[0]:.2e} new {res[1]:.2e}", flush=True)
