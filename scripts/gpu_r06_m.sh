#!/bin/bash
# round 6, m: the payload stage (guber_wire_pool_*) on the GPU — its tests, then tools/bench_pool_c api = wire beside api = c
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wire_pool.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5 > $O/tests.txt
cat $O/tests.txt
K=10000000
for cfg in "64 8" "256 8" "128 8" "64 1" "64 12" "16 8"; do
  set -- $cfg
  timeout 120 tools/bench_pool_c $1 $2 1000 $K 2.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
done
timeout 120 tools/bench_pool_c 64 8 1000 $K 2.0 200 c 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
timeout 120 tools/bench_pool_c 16 8 1 $K 1.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
timeout 120 tools/bench_pool_c 1 8 1 $K 1.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
cat $O/pool_wire.txt
