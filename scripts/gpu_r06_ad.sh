#!/bin/bash
# round 6, ad: the payload stage with its final defaults (twelve stages of 49 152 items): the bench's cases twice, the C api beside them; tests
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_ad; mkdir -p $O; : > $O/ad.txt
K=10000000
run() { r=$(timeout 120 tools/bench_pool_c $1 $2 $3 $K 2.0 200 $4 2>&1 | grep "^pool:\|^wire pool" | sed -n 's/.*keys: *\([0-9.]*\) M decisions.*p50 \([0-9.]*\) us p99 \([0-9.]*\) us, conservation: [0-9]* keys [0-9]* decisions \([0-9]*\) violations.*/\1 M\/s p50 \2 p99 \3 violations \4/p; s/wire pool: \([0-9]*\) stages (left because: \([^)]*\)).* \([0-9.]*\) items, .*/   \1 stages (\2) \3 items/p' | tr '\n' ' '); echo "api $4: $1 callers x $3-item RPCs, $2 tables: $r" | tee -a $O/ad.txt; }
for rep in 1 2; do
  for T in 64 128 192 256 320; do run $T 8 1000 wire; done
  run 64 12 1000 wire; run 192 12 1000 wire; run 256 1 1000 wire; run 1 8 1 wire
  run 64 8 1000 c; run 192 8 1000 c; run 256 8 1000 c
done
timeout 1200 python -m pytest tests/test_gpu_front.py tests/test_gpu_wire_pool.py tests/test_gpu_wire_dev.py tests/test_gpu_host_layer.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2 | tee $O/tests.txt
