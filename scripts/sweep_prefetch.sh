#!/bin/bash
# Pipelined step against --prefetch-depth and --persist-grid on ONE box, interleaved: bash scripts/sweep_prefetch.sh
set -e
mkdir -p gpurun_out
out=gpurun_out/sweep_prefetch.txt; : > $out
run() {
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-decoder-batch256 --no-alt-mode "$@" > gpurun_out/_sp.log 2>gpurun_out/_sp.err
  python3 - "$*" <<'PY' >> gpurun_out/sweep_prefetch.txt
import json, sys
d = json.loads([l for l in open("gpurun_out/_sp.log") if l.startswith("{")][-1])
print(f"{sys.argv[1]:40s} {d['value']:8.1f} img/s {d['ms_per_step']:.3f} ms dropped {d.get('prefetch_dropped')}")
PY
}
for r in 1 2; do
  for d in 2 3 4 6; do run --prefetch-depth $d; done
  for g in 208 224 256; do run --persist-grid $g; done
done
cat $out
