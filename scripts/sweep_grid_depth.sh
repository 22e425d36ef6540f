#!/bin/bash
# bench.py over (persistent grid, forwards in flight) pairs: "G:D G:D ..." (G = 0: library default)
# usage: bash scripts/sweep_grid_depth.sh <out.log> "224:2 49:5 ..." [extra bench flags]
OUT=$1; shift
PAIRS=$1; shift
: > $OUT
for p in $PAIRS; do
  G=${p%%:*}; D=${p##*:}
  LINE=$(python3 bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-alt-mode --persist-grid $G --prefetch-depth $D "$@" 2>>$OUT.err | tail -1)
  echo "$p $(echo "$LINE" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["stages_ms"]["resnet152_fwd"])')" >> $OUT
done
cat $OUT
