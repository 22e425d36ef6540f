#!/bin/bash
# Runs on the GPU box (via gpurun): bench lines, rocprofv3 kernel stats and the four PMC passes for one round tag.
# usage: bash scripts/collect_profiles.sh <tag>     (outputs under gpurun_out/<tag>/)
set -e
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-mode --no-overlap > $OUT/bench_n1_no_overlap.json 2>> $OUT/bench_n1.err
python3 $R/bench.py --steps 5 --warmup 2 --batch 256 --no-cpu-baseline > $OUT/bench_n1_batch256.json 2>> $OUT/bench_n1.err
python3 $R/bench.py --steps 20 --warmup 5 --batch 32 --no-cpu-baseline --no-decoder-batch256 > $OUT/bench_n1_batch32.json 2>> $OUT/bench_n1.err      # config 3's per-rank shape
python3 $R/bench.py --steps 10 --warmup 3 --hard --no-decoder-batch256 > $OUT/bench_hard_n1.json 2>> $OUT/bench_n1.err
echo "bench lines done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-alt-mode --no-decoder-batch256 > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_no_overlap -o run -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-alt-mode --no-decoder-batch256 --no-overlap > /dev/null 2>> $OUT/rocprof.err
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
  NAME=$(echo $SET | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pmc_$NAME -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-mode --no-decoder-batch256 --no-overlap > /dev/null 2>> $OUT/rocprof.err
  echo "pmc $NAME done"
done
python3 $R/scripts/pmc_summary.py $OUT $OUT/pmc_per_kernel.json > $OUT/pmc_summary.txt
# the same four passes at batch 256 (the configuration the MFMA-utilisation target is quoted on): bench.py reports counter
# values only from a summary collected at the batch it runs (profiles/<tag>_pmc_per_kernel_batch256.json)
mkdir -p $OUT/b256
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
  NAME=$(echo $SET | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/b256/pmc_$NAME -o run -- python3 $R/bench.py --steps 2 --warmup 1 --batch 256 --no-cpu-baseline --no-alt-mode --no-decoder-batch256 --no-overlap > /dev/null 2>> $OUT/rocprof.err
  echo "pmc batch256 $NAME done"
done
python3 $R/scripts/pmc_summary.py $OUT/b256 $OUT/pmc_per_kernel_batch256.json > $OUT/pmc_summary_batch256.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_batch256 -o run -- python3 $R/bench.py --steps 4 --warmup 2 --batch 256 --no-cpu-baseline --no-alt-mode --no-decoder-batch256 --no-overlap > /dev/null 2>> $OUT/rocprof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_dpt -o run -- python3 $R/bench.py --dpt --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_dpt_under_rocprof.json 2>> $OUT/rocprof.err
python3 $R/bench.py --dpt --steps 10 --warmup 3 > $OUT/bench_dpt.json 2>> $OUT/bench_n1.err
echo collected
