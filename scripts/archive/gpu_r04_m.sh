#!/bin/bash
# round 4, step m: device routing with its inputs staged through HBM: the routing test, the pool A/B, a kernel trace of the pool under load
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; O=gpurun_out/r04_m
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_layer.py -m gpu -q -x -k "rout or pool or host" > ${O}_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" ${O}_pytest_gpu.txt | cut -c1-300
{
for dr in 1 0; do
  for cfg in "64 8 1000" "256 8 1000" "16 8 1" "64 8 100"; do
    set -- $cfg
    echo "GUBER_POOL_DEVROUTE=$dr"
    GUBER_POOL_DEVROUTE=$dr GUBER_POOL_DEBUG=1 timeout 120 tools/bench_pool_c $1 $2 $3 10000000 2.0 200 2>&1 | grep -v amdgpu.ids
  done
done
} 2>&1 | tee ${O}_pool_devroute.txt | cut -c1-420
cd /tmp; export TMPDIR=/tmp
for dr in 1 0; do
  GUBER_POOL_DEVROUTE=$dr timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tr$dr -o t -- $R/tools/bench_pool_c 64 8 1000 10000000 1.0 200 > /tmp/tr$dr.log 2>&1
  f=$(find /tmp/tr$dr -name '*kernel_stats.csv' | head -1); echo "== DEVROUTE=$dr $f"; [ -n "$f" ] && cp $f $R/${O}_trace_devroute${dr}_kernel_stats.csv && head -12 $f | cut -c1-200
  tail -2 /tmp/tr$dr.log | cut -c1-300
done
