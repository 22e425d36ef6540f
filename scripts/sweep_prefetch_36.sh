#!/bin/bash
# forwards in flight 3 vs 6 at batch 32 and 64, three interleaved rounds on one box.   bash scripts/sweep_prefetch_36.sh
mkdir -p gpurun_out
out=gpurun_out/sweep_prefetch_36.txt; : > $out
run() {
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-decoder-batch256 --no-alt-mode "$@" > gpurun_out/_sc.log 2>gpurun_out/_sc.err
  python3 - "$*" <<'PY' >> gpurun_out/sweep_prefetch_36.txt
import json, sys
d = json.loads([l for l in open("gpurun_out/_sc.log") if l.startswith("{")][-1])
print(f"{sys.argv[1]:36s} {d['value']:8.1f} img/s {d['ms_per_step']:.3f} ms")
PY
}
for r in 1 2 3; do for b in 32 64; do for d in 3 6; do run --batch $b --prefetch-depth $d; done; done; done
cat $out
