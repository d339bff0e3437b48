#!/bin/bash
# the pool: who routes (callers / device) x callers x generations in flight, now that admission is first come first served
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=$R/gpurun_out/r05_j; mkdir -p $O
for dr in 0 1; do for depth in 2 3; do for callers in 64 128 256; do
  echo "DEVROUTE=$dr DEPTH=$depth callers=$callers"
  GUBER_POOL_DEVROUTE=$dr GUBER_POOL_DEPTH=$depth timeout 60 tools/bench_pool_c $callers 8 1000 10000000 2.0 200 2>&1 | grep -v amdgpu.ids | cut -c1-330
done; done; done | tee $O/pool_matrix.txt
echo "one table:"; for callers in 64 128 256; do timeout 60 tools/bench_pool_c $callers 1 1000 10000000 2.0 200 2>&1 | grep -v amdgpu.ids | cut -c1-330; done | tee -a $O/pool_matrix.txt
