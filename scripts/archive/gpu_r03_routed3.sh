#!/bin/bash
# small RPCs on the routed pool (one-launch path for a front stage of <= 256 requests) + parity of both routed paths
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_layer.py -m gpu -q -x 2>&1 | tail -15 | cut -c1-250
run() { timeout 120 tools/bench_pool_c $1 $2 $3 10000000 2.0 200 c 2>&1 | grep -v amdgpu.ids | cut -c1-330; }
{
run 1 8 1; run 4 8 1; run 16 8 1; run 64 8 1; run 16 8 10; run 64 8 100; run 64 8 1000
echo "== GUBER_POOL_ROUTED=0"; GUBER_POOL_ROUTED=0 run 16 8 1; GUBER_POOL_ROUTED=0 run 64 8 1
} 2>&1 | tee gpurun_out/r03_routed_small.txt
