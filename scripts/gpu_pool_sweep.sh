#!/bin/bash
# pool surface sweep on the GPU box: caller threads x shards at 10 M keys (tools/bench_pool.cpp), plus 1-item RPCs
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03_pool}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for cfg in "64 8 1000" "128 8 1000" "256 8 1000" "128 12 1000" "256 12 1000" "128 4 1000" "128 1 1000" "16 8 1" "256 8 1" "64 8 100"; do
  set -- $cfg
  timeout 120 tools/bench_pool_c $1 $2 $3 10000000 2.0 2>&1 | grep -v amdgpu.ids
done | tee $O/pool_sweep.txt
