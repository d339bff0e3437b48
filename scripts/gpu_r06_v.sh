#!/bin/bash
# round 6, v: the payload stage after the decoder stopped taking its engine's lock; the answers' last hop on the routing stream (lab knob) beside it
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_v; mkdir -p $O
K=10000000
for cfg in "64 8" "128 8" "256 8" "256 1"; do
  set -- $cfg
  timeout 120 tools/bench_pool_c $1 $2 1000 $K 2.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
done
mkdir -p /tmp/lablib; cp gubernator_amd/libguber_hip_lab.so /tmp/lablib/libguber_hip.so
echo "--- GUBER_FRONT_OUT_ON_EVAL=0 (laboratory build)" >> $O/pool_wire.txt
for cfg in "64 8" "128 8" "256 8"; do
  set -- $cfg
  GUBER_FRONT_OUT_ON_EVAL=0 LD_LIBRARY_PATH=/tmp/lablib timeout 120 tools/bench_pool_c $1 $2 1000 $K 2.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
done
echo "--- two engine streams" >> $O/pool_wire.txt
GUBER_BENCH_WIRE_ENGINE_STREAMS=2 timeout 120 tools/bench_pool_c 256 8 1000 $K 2.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
echo "--- api c" >> $O/pool_wire.txt
timeout 120 tools/bench_pool_c 64 8 1000 $K 2.0 200 c 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
timeout 120 tools/bench_pool_c 256 8 1000 $K 2.0 200 c 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
cat $O/pool_wire.txt
