#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r02_ah; mkdir -p $O; : > $O/pool.txt
timeout 300 python -m pytest tests/test_gpu_host_layer.py -m gpu -q -x > $O/pytest_host.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_host.txt | cut -c1-300
for idle in 0 20; do
  for cfg in "64 4 1000" "64 1 1000" "256 4 1" "16 4 1"; do
    echo -n "idle_us=$idle  " | tee -a $O/pool.txt; GUBER_POOL_IDLE_US=$idle timeout 60 tools/bench_pool_c $cfg 1000000 1 | tee -a $O/pool.txt
  done
done
