#!/bin/bash
# rocprofv3 kernel trace of the un-overlapped bench; prints per-step time of every (kernel, grid) that is not part of the ResNet forward
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ms_stats
rocprofv3 --kernel-trace --output-format csv -d /tmp/ms_stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-mode --no-decoder-batch256 --no-overlap > /tmp/ms_stats.log 2>&1
f=$(find /tmp/ms_stats -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/main_stream_stats.txt
import collections,csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
steps=15.0   # 3 warm-up + 10 timed + stage-timed + profiled
skip=("persist_ws","halo","bn_apply","bn_finalize","bn_stats_slice","tail_fixup","stem_pack","ws256","poison","clear_status")
agg=collections.OrderedDict()
for r in rows:
    n=r["Kernel_Name"]
    if any(s in n for s in skip): continue
    k=(n[:100], int(r["Grid_Size_X"])//max(1,int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"])//max(1,int(r["Workgroup_Size_Y"])))
    agg.setdefault(k,[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
tot=0
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    t=sum(v)/steps; tot+=t
    if t>=3: print(f"{t:8.1f} us/step  {len(v)/steps:6.1f} calls/step  avg {sum(v)/len(v):7.1f} us  grid {k[1]}x{k[2]}  {k[0]}")
print(f"total non-ResNet kernel time per step: {tot:.1f} us")
PY
