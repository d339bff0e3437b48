#!/bin/bash
# round 6, at: the payload stage's defaults once more, now that the callers no longer write the varints (a caller costs its CPU less: do larger / fewer / more stages pay?)
# stages x items per stage at 128 / 192 / 256 callers, 8 tables, alternating, two repetitions; every run gated by conservation
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_at; mkdir -p $O; : > $O/at.txt
K=10000000
for rep in 1 2; do for T in 128 192 256; do for cfg in "12 49152" "12 32768" "12 65536" "8 49152" "8 65536" "6 98304"; do
  set -- $cfg
  r=$(GUBER_BENCH_WIRE_STAGES=$1 GUBER_BENCH_WIRE_ITEMS=$2 timeout 120 tools/bench_pool_c $T 8 1000 $K 2.0 200 wire 2>&1 | grep "^pool:" | sed -n 's/.*keys: *\([0-9.]*\) M decisions.*p50 \([0-9.]*\) us p99 \([0-9.]*\) us, conservation: [0-9]* keys [0-9]* decisions \([0-9]*\) violations.*/\1 M\/s p50 \2 p99 \3 violations \4/p')
  echo "rep $rep $T callers, $1 stages of $2 items: $r" | tee -a $O/at.txt
done; done; done
