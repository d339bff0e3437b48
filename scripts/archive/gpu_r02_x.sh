#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r02_x; mkdir -p $O
for cfg in "8 4 1000" "32 4 1000" "64 4 1000" "128 4 1000" "64 1 1000" "64 8 1000" "64 4 100" "256 4 1"; do
  timeout 120 tools/bench_pool_c $cfg 1000000 2 | tee -a $O/pool.txt
done
