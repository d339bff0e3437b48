#!/bin/bash
# round 6, ag: does counting the records of every payload (up to 256 KB; before: up to 8 KB) cost the callers anything that shows?  the product against a
# measurement build with the old threshold, alternating on one box
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_ag; mkdir -p $O; : > $O/ag.txt
K=10000000
mkdir -p /tmp/v8k; cp gubernator_amd/libguber_hip_v_walk8k.so /tmp/v8k/libguber_hip.so
for rep in 1 2 3; do for T in 64 192 256; do for v in product walk8k; do
  if [ $v = product ]; then L=""; else L=/tmp/v8k; fi
  r=$(LD_LIBRARY_PATH=$L timeout 120 tools/bench_pool_c $T 8 1000 $K 2.0 200 wire 2>&1 | grep "^pool:" | sed -n 's/.*keys: *\([0-9.]*\) M decisions.*p50 \([0-9.]*\) us p99 \([0-9.]*\) us, conservation: [0-9]* keys [0-9]* decisions \([0-9]*\) violations.*/\1 M\/s p50 \2 p99 \3 violations \4/p')
  echo "rep $rep callers $T $v: $r" | tee -a $O/ag.txt
done; done; done
