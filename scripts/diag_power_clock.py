"""Sample GPU power / clock from sysfs (hwmon, no privileges) while a workload runs in this process:
  python scripts/diag_power_clock.py step        - the pipelined training step (bench shape)
  python scripts/diag_power_clock.py forward N G  - N ResNet forwards at once, persistent grid G
Prints cap, mean / max power and the shader clock readings seen during the timed loop."""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from depth_image_captioning_pub_amd import native, synthetic as syn, _lib
from depth_image_captioning_pub_amd.engine import CaptionTrainer

hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
print("hwmon:", hw)
for h in hw[:1]:
    print({os.path.basename(f): open(f).read().strip() for f in sorted(glob.glob(h + "/*")) if os.path.isfile(f) and os.access(f, os.R_OK) and os.path.getsize(f) <= 4096 and not f.endswith("uevent")})


def read(path):
    try:
        return int(open(path).read().split()[0])
    except Exception:
        return None


samples, stop = [], False
allcards = []      # (t, [power per card], [sclk per card]): the card this process runs on is the one whose power moves
def sampler():
    while not stop and hw:
        t = time.perf_counter()
        allcards.append((t, [read(h + "/power1_input") or 0 for h in hw], [read(h + "/freq1_input") or 0 for h in hw]))
        time.sleep(0.002)


DEV = "cuda:0"
CONV_MODE = os.environ.get("DIC_CONV_MODE", "f16x2")      # ResNet arithmetic of the run (f16x2 = bench default since round 3)
mode = sys.argv[1] if len(sys.argv) > 1 else "step"
if mode == "step":
    tr = CaptionTrainer(10000, device=DEV, seed=123, conv_mode=CONV_MODE)
    imgs = syn.rgb_images(64, seed=123).to(DEV); depth = syn.depth_maps(64, seed=123).to(DEV)
    caps, lens = syn.captions_fixed(64, 10000, 20, seed=123); caps = caps.to(DEV)
    body = lambda: tr.train_step(imgs, depth, caps, lens, next_imgs=[imgs] * tr.prefetch_depth)
    iters = 150
else:
    n, G = int(sys.argv[2]), int(sys.argv[3])
    _lib.check(_lib.load().dic_conv_persistent_grid(G))
    rn = {k: v.to(DEV) for k, v in syn.resnet152_weights(seed=125).items()}
    stat = lambda k: k.endswith("running_mean") or k.endswith("running_var")
    runners = [native.ResNetRunner({k: (v.clone() if stat(k) else v) for k, v in rn.items()}, conv_mode=CONV_MODE) for _ in range(n)]
    imgs = syn.rgb_images(64, seed=123).to(DEV)
    outs = [torch.empty((64, 49, 2048), device=DEV) for _ in range(n)]
    streams = [torch.cuda.Stream() for _ in range(n)]
    def body():
        for i in range(n):
            with torch.cuda.stream(streams[i]):
                runners[i].forward(imgs, True, out=outs[i], compact=True)
    iters = 40
for _ in range(5):
    body()
torch.cuda.synchronize()
th = threading.Thread(target=sampler); th.start()
time.sleep(0.3)
t0 = time.perf_counter()
for _ in range(iters):
    body()
torch.cuda.synchronize()
t1 = time.perf_counter()
time.sleep(0.3)
stop = True; th.join()
import statistics
nb = [i for i, (t, _, _) in enumerate(allcards) if t0 <= t <= t1]
ni = [i for i, (t, _, _) in enumerate(allcards) if t < t0 - 0.05 or t > t1 + 0.2]
delta = [statistics.mean(allcards[i][1][c] for i in nb) - statistics.mean(allcards[i][1][c] for i in ni) for c in range(len(hw))] if nb and ni else []
print("mean power busy - idle per card (W):", [round(d / 1e6) for d in delta])
c = max(range(len(hw)), key=lambda i: delta[i]) if delta else 0
print("card under load:", hw[c])
busy = [(allcards[i][1][c], allcards[i][2][c]) for i in nb]
idle = [(allcards[i][1][c], allcards[i][2][c]) for i in ni]
if busy:
    ps = [p for p, _ in busy]; fs = [f for _, f in busy if f]
    print(f"{mode}: {(t1 - t0) / iters * 1e3:.2f} ms per iteration; power W mean {sum(ps) / len(ps) / 1e6:.0f} max {max(ps) / 1e6:.0f} "
          f"(idle {sum(p for p, _ in idle) / max(1, len(idle)) / 1e6:.0f}); sclk MHz mean {sum(fs) / max(1, len(fs)) / 1e6:.0f} min {min(fs) / 1e6 if fs else 0:.0f} max {max(fs) / 1e6 if fs else 0:.0f}; {len(busy)} samples")
else:
    print("no power samples readable", len(samples))
