#!/bin/bash
# timing-only ablations of the on-the-fly-operand kernel inside a real ResNet forward (experiments library; results are wrong under 57..59):
# which part of its 55 us is the residual read / the fp32 copy it writes.   scripts/ablate_ws6.sh
export DIC_LIB=experiments
for codes in "" 57 58 59; do
  echo "== switches: ${codes:-none}"
  bash $GRAFT_REPO_ROOT/scripts/trace_fwd.sh abl_${codes:-none} "$codes" | grep -E "ws6|halo|ws256|kernel time|last forward"
done
