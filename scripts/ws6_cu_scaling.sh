#!/bin/bash
# Is the on-the-fly 1x1 kernel bound per CU or chip-wide?  Same kernel, cold operands (--rotate 4), 196 / 224 / 256 tiles on as many
# workgroups (experiments library: switch 82 = persistent grids of up to 256).   bash scripts/ws6_cu_scaling.sh
export DIC_LIB=experiments
for b in 64 73 83; do
  echo "== batch $b (layer3 conv1: $((b*196)) rows, $(( (b*196+127)/128*2 )) tiles)"
  python3 scripts/bench_conv1x1_bn.py --batch $b --fmt 1 --rotate 4 --switches 82 --iters 40 --only "layer3 conv1"
done
