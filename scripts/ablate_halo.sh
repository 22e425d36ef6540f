#!/bin/bash
# timing-only ablations of the LDS-halo kernel (on-the-fly form) inside a real ResNet forward (experiments library; results are wrong
# under 64..69): how much of its ~53 us is the weight stream / the input loads + transform.   scripts/ablate_halo.sh
export DIC_LIB=experiments
for codes in ${HALO_CODES:-0 64 65 66 67 68 69}; do      # 0 = none
  [ "$codes" = 0 ] && codes=""
  echo "== switches: ${codes:-none}"
  bash $GRAFT_REPO_ROOT/scripts/trace_fwd.sh habl_${codes:-none} "$codes" | grep -E "ws6|halo|ws256|kernel time|last forward"
done
