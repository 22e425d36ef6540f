#!/bin/bash
# timing-only ablations of the 256x128 f16x2 kernel (conv3 of every block) inside a real ResNet forward (experiments library; results are
# wrong under 44..47): 44 = no DMA in the loop, 45 = no tile stores, 46 = neither, 47 = no fragment reads.   scripts/ablate_ws256.sh
export DIC_LIB=experiments
for codes in ${WS256_CODES:-0 44 45 46 47}; do
  [ "$codes" = 0 ] && codes=""
  echo "== switches: ${codes:-none}"
  bash $GRAFT_REPO_ROOT/scripts/trace_fwd.sh w256_${codes:-none} "$codes" | grep -E "ws6|halo|ws256|kernel time|last forward"
done
