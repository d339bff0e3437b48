#!/bin/bash
# round 6, z: a soak of the payload stage on the GPU — tools/bench_pool_c api = wire over caller counts, tables, RPC sizes, stage counts and sizes (small
# stages: callers close full ones all the time), every run gated by per-key conservation over all its answers; then the C99 twin of the Go handlers and the
# wire files of the GPU suite
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_z; mkdir -p $O; : > $O/soak.txt
n=0; bad=0
for T in 3 17 64 200; do for S in 1 3 8 12; do for items in 1 7 100 1000; do
  case $(( (T + S + items) % 3 )) in 0) st=2; mi=4096; dq=1;; 1) st=3; mi=16384; dq=2;; *) st=6; mi=131072; dq=2;; esac
  [ $items -gt $mi ] && mi=131072
  out=$(GUBER_BENCH_WIRE_STAGES=$st GUBER_BENCH_WIRE_ITEMS=$mi GUBER_BENCH_WIRE_DECODES=$dq timeout 120 tools/bench_pool_c $T $S $items 200000 0.25 150 wire 2>&1 | grep "^pool:")
  n=$((n+1))
  v=$(echo "$out" | sed -n 's/.*conservation: \([0-9]*\) keys \([0-9]*\) decisions \([0-9]*\) violations.*/\3/p')
  e=$(echo "$out" | sed -n 's/.*errors \([0-9]*\),.*/\1/p')
  r=$(echo "$out" | sed -n 's/.*keys: *\([0-9.]*\) M decisions.*/\1/p')
  echo "T $T S $S items $items stages $st max_items $mi decodes $dq: $r M/s errors ${e:-?} violations ${v:-?}" >> $O/soak.txt
  if [ "${v:-x}" != "0" ] || [ "${e:-x}" != "0" ]; then bad=$((bad+1)); echo "$out" >> $O/soak.txt; fi
done; done; done
echo "soak: $n runs, $bad bad" | tee -a $O/soak.txt
tail -5 $O/soak.txt
timeout 900 python -m pytest tests/test_gpu_host_layer.py -m gpu -x -q -p no:cacheprovider -k plain_c 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_wire_pool.py tests/test_gpu_wire_dev.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
