#!/bin/bash
# round 6, aa: the payload stage's generations as ONE pair of launches for all tables (launch_group_mem) instead of the owner-partitioned pipeline in groups of four
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_aa; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_front.py tests/test_gpu_wire_pool.py tests/test_gpu_wire_dev.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 | tee $O/tests.txt
K=10000000
mkdir -p /tmp/lablib; cp gubernator_amd/libguber_hip_lab.so /tmp/lablib/libguber_hip.so
for cfg in "64 8" "128 8" "256 8" "64 12" "256 1"; do
  set -- $cfg
  timeout 120 tools/bench_pool_c $1 $2 1000 $K 2.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
  echo "--- the same with GUBER_FRONT_ONE_PAIR_MAX=0 (laboratory build: the owner-partitioned pipeline in groups of four)" >> $O/pool_wire.txt
  GUBER_FRONT_ONE_PAIR_MAX=0 LD_LIBRARY_PATH=/tmp/lablib timeout 120 tools/bench_pool_c $1 $2 1000 $K 2.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
done
timeout 120 tools/bench_pool_c 1 8 1 $K 1.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
cat $O/pool_wire.txt
