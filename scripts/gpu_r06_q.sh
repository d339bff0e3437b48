#!/bin/bash
# round 6, q: the payload stage — sleepers woken by the stage's capacity, a flusher with priority
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_q; mkdir -p $O
K=10000000
for cfg in "128 8" "256 8" "512 8" "1024 8" "256 1"; do
  set -- $cfg
  timeout 120 tools/bench_pool_c $1 $2 1000 $K 2.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
done
GUBER_BENCH_WIRE_DECODES=3 timeout 120 tools/bench_pool_c 512 8 1000 $K 2.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
GUBER_BENCH_WIRE_ITEMS=65536 timeout 120 tools/bench_pool_c 512 8 1000 $K 2.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
GUBER_BENCH_WIRE_ITEMS=262144 GUBER_BENCH_WIRE_STAGES=8 timeout 120 tools/bench_pool_c 512 8 1000 $K 2.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
cat $O/pool_wire.txt
