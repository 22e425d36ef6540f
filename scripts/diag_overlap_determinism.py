"""Are the ResNet forward and the rest of the step bit-reproducible when they run CONCURRENTLY (two streams)?  Replays the
same ResNet forward on a side stream while the main stream repeats the same gradient computation (fixed weights), and
compares every repetition with the first one bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_image_captioning_pub_amd import synthetic as syn
from depth_image_captioning_pub_amd.engine import CaptionTrainer

dev, B, V, T = "cuda:0", 64, 10000, 20
tr = CaptionTrainer(V, device=dev, seed=123, conv_mode=os.environ.get("DIC_CONV_MODE", "bf16x3"))      # DIC_CONV_MODE=f16x2: the default arithmetic
for code in sys.argv[3:]:
    if code == "nocompact":
        tr.compact_ok = False
    else:
        from depth_image_captioning_pub_amd import _lib
        _lib.load().dic_debug_force_staged_gemm(int(code))
imgs = syn.rgb_images(B, seed=123).to(dev); depth = syn.depth_maps(B, seed=123).to(dev)
caps, lens = syn.captions_fixed(B, V, T, seed=123); caps = caps.to(dev)
drop = syn.dropout_multiplier(B, T, 0.5, seed=123).to(dev)
feats0 = tr.resnet.forward(imgs, True, compact=tr.compact_ok).clone()
torch.cuda.synchronize()
concurrent = len(sys.argv) < 2 or sys.argv[1] != "serial"
ref_f = ref_g = ref_l = None
bad_f = bad_g = bad_l = 0
seen = set()
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for it in range(N):
    if concurrent:
        tr.prefetch_features(imgs, compact=True)            # side stream: the same forward again
    loss = tr.train_step(None, depth, caps, lens, drop_mult=drop, precomputed_features=feats0, apply_update=False)
    g = tr.flat.grad.clone()
    if concurrent:
        f = tr._take_prefetched(imgs).clone()
    else:
        f = tr.resnet.forward(imgs, True, compact=True).clone()
    torch.cuda.synchronize()
    l = float(loss.item())
    if ref_f is None:
        ref_f, ref_g, ref_l = f, g, l
    else:
        bad_f += int(not torch.equal(f, ref_f)); bad_g += int(not torch.equal(g, ref_g)); bad_l += int(l != ref_l)
        if not torch.equal(g, ref_g):
            d = (g - ref_g).abs()
            # which tensor?
            names = [k for k in tr.flat.names if float(tr.flat.view(d, k).max()) > 0]
            seen.update(names)
print("tensors that differed:", sorted(seen))
print(f"{' '.join(sys.argv[1:])}: {N - 1} repeats; features differ {bad_f}, gradients differ {bad_g}, loss differs {bad_l}")
