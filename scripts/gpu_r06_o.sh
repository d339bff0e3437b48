#!/bin/bash
# round 6, o: the payload stage after the decode's command diet, two decodes queued, the routing ahead of the evaluation
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wire_pool.py tests/test_gpu_wire_dev.py tests/test_gpu_front.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5 > $O/tests.txt
cat $O/tests.txt
K=10000000
for cfg in "64 8" "128 8" "256 8" "512 8" "64 1" "256 1"; do
  set -- $cfg
  timeout 120 tools/bench_pool_c $1 $2 1000 $K 2.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
done
GUBER_BENCH_WIRE_DECODES=1 timeout 120 tools/bench_pool_c 64 8 1000 $K 2.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
GUBER_BENCH_WIRE_DECODES=3 timeout 120 tools/bench_pool_c 256 8 1000 $K 2.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
timeout 120 tools/bench_pool_c 1 8 1 $K 1.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
cat $O/pool_wire.txt
