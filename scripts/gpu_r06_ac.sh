#!/bin/bash
# round 6, ac: stages capped at the one-pair threshold (49 152 items) with more of them in rotation, at 192 ... 384 callers
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_ac; mkdir -p $O; : > $O/ac.txt
K=10000000
run() { r=$(env "$@" timeout 120 tools/bench_pool_c $T 8 1000 $K 2.0 200 wire 2>&1 | grep "^pool:\|^wire pool" | sed -n 's/.*keys: *\([0-9.]*\) M decisions.*p50 \([0-9.]*\) us p99 \([0-9.]*\) us, conservation: [0-9]* keys [0-9]* decisions \([0-9]*\) violations.*/\1 M\/s p50 \2 p99 \3 violations \4/p; s/wire pool: \([0-9]*\) stages (left because: \([^)]*\)).* \([0-9.]*\) items, .*/   \1 stages (\2) \3 items/p' | tr '\n' ' '); echo "callers $T $*: $r" | tee -a $O/ac.txt; }
for rep in 1 2; do for T in 192 256 384; do
  run GUBER_BENCH_WIRE_ITEMS=131072 GUBER_BENCH_WIRE_STAGES=6
  run GUBER_BENCH_WIRE_ITEMS=49152 GUBER_BENCH_WIRE_STAGES=8
  run GUBER_BENCH_WIRE_ITEMS=49152 GUBER_BENCH_WIRE_STAGES=12
  run GUBER_BENCH_WIRE_ITEMS=32768 GUBER_BENCH_WIRE_STAGES=12
done; done
