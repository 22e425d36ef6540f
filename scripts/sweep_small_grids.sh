#!/bin/bash
# Pipelined step with the persistent kernels on fewer, fatter workgroups (two / three / four 196-tile rounds per workgroup), interleaved
# with the default on one box.   bash scripts/sweep_small_grids.sh
mkdir -p gpurun_out
out=gpurun_out/sweep_small_grids.txt; : > $out
run() {
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-decoder-batch256 --no-alt-mode "$@" > gpurun_out/_sg.log 2>gpurun_out/_sg.err
  python3 - "$*" <<'PY' >> gpurun_out/sweep_small_grids.txt
import json, sys
d = json.loads([l for l in open("gpurun_out/_sg.log") if l.startswith("{")][-1])
print(f"{sys.argv[1]:40s} {d['value']:8.1f} img/s {d['ms_per_step']:.3f} ms; resnet alone {d['stages_ms'].get('resnet152_fwd')} ms")
PY
}
for r in 1 2; do
  for g in 224 98 66 49; do run --persist-grid $g; done
  run --persist-grid 98 --prefetch-depth 4
  run --persist-grid 98 --prefetch-depth 6
done
cat $out
