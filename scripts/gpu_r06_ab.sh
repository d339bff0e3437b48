#!/bin/bash
# round 6, ab: one pair of launches against the owner-partitioned groups at 192 / 256 callers, alternating on one box (laboratory build both times)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_ab; mkdir -p $O; : > $O/ab.txt
K=10000000
mkdir -p /tmp/lablib; cp gubernator_amd/libguber_hip_lab.so /tmp/lablib/libguber_hip.so
for rep in 1 2 3; do for T in 192 256; do for v in 131072 0; do
  r=$(GUBER_FRONT_ONE_PAIR_MAX=$v LD_LIBRARY_PATH=/tmp/lablib timeout 120 tools/bench_pool_c $T 8 1000 $K 2.0 200 wire 2>&1 | grep "^pool:" | sed -n 's/.*keys: *\([0-9.]*\) M decisions.*p50 \([0-9.]*\) us p99 \([0-9.]*\) us, conservation: [0-9]* keys [0-9]* decisions \([0-9]*\) violations.*/\1 M\/s p50 \2 p99 \3 violations \4/p')
  echo "rep $rep callers $T one_pair_max $v: $r" | tee -a $O/ab.txt
done; done; done
