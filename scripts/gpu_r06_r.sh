#!/bin/bash
# round 6, t: the payload stage with two threads of its own (intake, front)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_t; mkdir -p $O
K=10000000
for cfg in "64 8" "128 8" "256 8" "512 8" "256 1"; do
  set -- $cfg
  timeout 120 tools/bench_pool_c $1 $2 1000 $K 2.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
done
GUBER_BENCH_WIRE_DECODES=3 timeout 120 tools/bench_pool_c 256 8 1000 $K 2.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
GUBER_BENCH_WIRE_DECODES=4 timeout 120 tools/bench_pool_c 256 8 1000 $K 2.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
GUBER_BENCH_WIRE_ENGINE_STREAMS=2 timeout 120 tools/bench_pool_c 256 8 1000 $K 2.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
GUBER_BENCH_WIRE_DECODES=3 timeout 120 tools/bench_pool_c 384 8 1000 $K 2.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
cat $O/pool_wire.txt
