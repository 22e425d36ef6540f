#!/bin/bash
# A/B of bench.py over debug-switch sets, interleaved in one gpurun call: scripts/ab_bench.sh "<name>=<codes>" ...   (codes: comma list or "-")
# prints value / ms_per_step / resnet stage per configuration and round
out=gpurun_out/ab_$(date +%s).log
for round in 1 2; do
  for cfg in "$@"; do
    name=${cfg%%=*}; codes=${cfg#*=}
    if [ "$codes" = "-" ]; then unset DIC_DEBUG_SWITCHES; else export DIC_DEBUG_SWITCHES=$codes; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-mode --no-decoder-batch256 ${AB_ARGS} 2>>$out.err | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$name round $round:', d['value'], 'img/s', d['ms_per_step'], 'ms; resnet alone', d['stages_ms'].get('resnet152_fwd'), 'ms', flush=True)
" | tee -a $out
  done
done
