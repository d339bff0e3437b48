#!/bin/bash
# round 6, av: up to how many callers inside the pool should a one-request RPC be evaluated by its caller?  GUBER_WIRE_DIRECT = that number (laboratory build), 8 tables and one
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_av; mkdir -p $O; : > $O/av.txt
K=10000000
mkdir -p /tmp/lablib; cp gubernator_amd/libguber_hip_lab.so /tmp/lablib/libguber_hip.so
for S in 8 1; do for T in 2 4 8 16 32 64 128; do for D in 0 1 4 8 16 64 1000; do
  r=$(GUBER_WIRE_DIRECT=$D LD_LIBRARY_PATH=/tmp/lablib timeout 120 tools/bench_pool_c $T $S 1 $K 0.7 150 wire 2>&1 | grep "^pool:" | sed -n 's/.*keys: *\([0-9.]*\) M decisions.*p50 \([0-9.]*\) us p99 \([0-9.]*\) us, conservation: [0-9]* keys [0-9]* decisions \([0-9]*\) violations.*/\1 M\/s p50 \2 p99 \3 violations \4/p')
  echo "direct up to $D inside; $S table(s), $T callers x 1-item RPCs: $r" | tee -a $O/av.txt
done; done; done
