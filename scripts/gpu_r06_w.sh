#!/bin/bash
# round 6, w: the payload stage — the answers' last hop on the decoder's stream (lab knob), one decode at a time at 256 callers
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_w; mkdir -p $O
K=10000000
mkdir -p /tmp/lablib; cp gubernator_amd/libguber_hip_lab.so /tmp/lablib/libguber_hip.so
for cfg in "128 8" "256 8"; do
  set -- $cfg
  timeout 120 tools/bench_pool_c $1 $2 1000 $K 2.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
  echo "--- GUBER_WIRE_OUT_ON_OWN=1 (laboratory build)" >> $O/pool_wire.txt
  GUBER_WIRE_OUT_ON_OWN=1 LD_LIBRARY_PATH=/tmp/lablib timeout 120 tools/bench_pool_c $1 $2 1000 $K 2.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
done
echo "--- decodes 1" >> $O/pool_wire.txt
GUBER_BENCH_WIRE_DECODES=1 timeout 120 tools/bench_pool_c 256 8 1000 $K 2.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
echo "--- 192 / 320 callers" >> $O/pool_wire.txt
timeout 120 tools/bench_pool_c 192 8 1000 $K 2.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
timeout 120 tools/bench_pool_c 320 8 1000 $K 2.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
cat $O/pool_wire.txt
