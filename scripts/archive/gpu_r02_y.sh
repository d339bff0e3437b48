#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r02_y; mkdir -p $O; : > $O/pool.txt
timeout 900 python -m pytest tests/test_gpu_host_layer.py -m gpu -q -x > $O/pytest_host.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_host.txt | cut -c1-600
for cfg in "8 4 1000" "32 4 1000" "64 4 1000" "128 4 1000" "256 4 1000" "64 1 1000" "128 8 1000" "64 4 100" "256 4 1" "16 4 1"; do
  timeout 120 tools/bench_pool_c $cfg 1000000 2 | tee -a $O/pool.txt
done
