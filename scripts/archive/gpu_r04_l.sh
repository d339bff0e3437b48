#!/bin/bash
# round 4, step l: the GPU suite on the current tree, then the pool with the device routing its front stage (guber_stage_route) vs the callers routing
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; O=gpurun_out/r04_l
timeout 1500 python -m pytest tests -m gpu -q -x > ${O}_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 ${O}_pytest_gpu.txt | cut -c1-300
{
for dr in 1 0; do
  for cfg in "64 8 1000" "256 8 1000" "64 12 1000" "16 8 1" "64 8 100"; do
    set -- $cfg
    echo "GUBER_POOL_DEVROUTE=$dr"
    GUBER_POOL_DEVROUTE=$dr GUBER_POOL_DEBUG=1 timeout 120 tools/bench_pool_c $1 $2 $3 10000000 2.0 200 2>&1 | grep -v amdgpu.ids
  done
done
timeout 120 tools/bench_pool_c 64 1 1000 10000000 2.0 200 2>&1 | grep -v amdgpu.ids
} 2>&1 | tee ${O}_pool_devroute.txt | cut -c1-420
