#!/bin/bash
# The on-the-fly 1x1 kernel (layer-3 conv1, 196 tiles, cold operands) on 196 / 98 / 66 / 49 workgroups: does a CU stream faster when fewer
# CUs share the memory system?   bash scripts/ws6_fewer_cus.sh
for g in 196 98 66 49; do
  echo "== persistent grid $g"
  python3 scripts/bench_conv1x1_bn.py --batch 64 --fmt 1 --rotate 4 --iters 40 --only "layer3 conv1" --persist-grid $g 2>/dev/null | grep "layer3 conv1"
done
