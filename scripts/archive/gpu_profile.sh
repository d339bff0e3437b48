#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats + PMC traffic counters (separate passes) for bench.py.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
# the default bench command (256 timed steps on 4 logical shards, then the 32-step profiling leg and the
# 200-batch latency leg, both one batch at a time), minus the CPU leg
ARGS="--no-cpu-baseline"
for algo in token leaky; do
  rm -rf $R/gpurun_out/prof_$algo
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$algo -o $algo -- python $R/bench.py $ARGS --algo $algo > $R/gpurun_out/prof_$algo.log 2>&1; echo "stats $algo rc=$?"
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$ctr
  timeout 600 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_$ctr -o pmc -- python $R/bench.py --steps 32 --warmup 4 --no-cpu-baseline --profile-steps 0 > $R/gpurun_out/pmc_$ctr.log 2>&1; echo "pmc $ctr rc=$?"
done
ls $R/gpurun_out/prof_token $R/gpurun_out/pmc_FETCH_SIZE
head -5 $R/gpurun_out/prof_token/token_kernel_stats.csv
head -3 $R/gpurun_out/pmc_FETCH_SIZE/*counter_collection.csv
