#!/bin/bash
# Copies the summaries of one collected set (gpurun_out/<tag>/, scripts/collect_profiles.sh) into profiles/ under the names the
# earlier rounds use.  usage: bash scripts/publish_profiles.sh <tag>
set -e
T=$1; S=gpurun_out/$T; D=profiles
cp $S/bench_n1.json $D/${T}_bench_n1.json
cp $S/bench_n1_no_overlap.json $D/${T}_bench_n1_no_overlap.json
cp $S/bench_n1_batch256.json $D/${T}_bench_n1_batch256.json
cp $S/bench_under_rocprof.json $D/${T}_bench_under_rocprof.json
cp $S/bench_dpt.json $D/${T}_bench_dpt.json
cp $S/stats/run_kernel_stats.csv $D/${T}_bench_kernel_stats.csv
cp $S/stats_no_overlap/run_kernel_stats.csv $D/${T}_bench_kernel_stats_no_overlap.csv
cp $S/stats_batch256/run_kernel_stats.csv $D/${T}_bench_kernel_stats_batch256_no_overlap.csv
cp $S/stats_dpt/run_kernel_stats.csv $D/${T}_dpt_kernel_stats.csv
cp $S/pmc_per_kernel.json $D/${T}_pmc_per_kernel.json
cp $S/pmc_per_kernel_batch256.json $D/${T}_pmc_per_kernel_batch256.json
ls -la $D | grep ${T}_
