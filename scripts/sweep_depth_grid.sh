#!/bin/bash
# bench.py over (forwards in flight, persistent grid): scripts/sweep_depth_grid.sh
for rep in 1 2; do
for cfg in "3 0" "2 0" "4 0" "3 196" "3 256" "3 208"; do
  set -- $cfg
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-mode --no-decoder-batch256 --prefetch-depth $1 --persist-grid $2 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('depth $1 grid $2 rep $rep:', d['value'], 'img/s', d['ms_per_step'], 'ms', flush=True)"
done
done
