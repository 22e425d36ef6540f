#!/bin/bash
# scripts/trace_fwd.sh <tag> [switch codes] : rocprofv3 kernel trace of ResNet forwards (f16x2), aggregated by kernel kind for the last forward
tag=$1; codes=$2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_$tag -- python3 $GRAFT_REPO_ROOT/scripts/run_resnet_fwd.py --conv-mode f16x2 ${codes:+--switches $codes} > /tmp/trace_$tag.log 2>&1
tail -1 /tmp/trace_$tag.log
python3 $GRAFT_REPO_ROOT/scripts/agg_trace.py /tmp/trace_$tag | tee $GRAFT_REPO_ROOT/gpurun_out/trace_$tag.txt
