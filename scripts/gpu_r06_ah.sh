#!/bin/bash
# round 6, ah: the decode's arguments and payload bytes in ONE copy (six stream commands per decode): tests, then the payload stage's cases twice
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_ah; mkdir -p $O; : > $O/ah.txt
timeout 1200 python -m pytest tests/test_gpu_wire_pool.py tests/test_gpu_wire_dev.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2 | tee $O/tests.txt
K=10000000
run() { r=$(timeout 120 tools/bench_pool_c $1 $2 $3 $K 2.0 200 $4 2>&1 | grep "^pool:\|^wire pool" | sed -n 's/.*keys: *\([0-9.]*\) M decisions.*p50 \([0-9.]*\) us p99 \([0-9.]*\) us, conservation: [0-9]* keys [0-9]* decisions \([0-9]*\) violations.*/\1 M\/s p50 \2 p99 \3 violations \4/p; s/wire pool: \([0-9]*\) stages (left because: \([^)]*\)).* sealed -> decoded \([0-9]*\) us, decoded -> answers in host memory \([0-9]*\) us, \([0-9.]*\) items, .*decode enqueue \([0-9.]*\) us.*/   \1 stages, decode \3 us, evaluate \4 us, \5 items, decode enqueue \6 us/p' | tr '\n' ' '); echo "api $4: $1 callers x $3-item RPCs, $2 tables: $r" | tee -a $O/ah.txt; }
for rep in 1 2; do
  for T in 64 128 192 256; do run $T 8 1000 wire; done
  run 256 1 1000 wire; run 1 8 1 wire
done
