#!/bin/bash
# round 6, au: the payload stage's direct path (a lone one-request RPC evaluated by its caller): the wire files of the GPU suite, then 1-item RPCs from 1 / 2 / 16 callers
# with and without it (laboratory build, GUBER_WIRE_DIRECT=0), 1 and 8 tables; 1000-item RPCs unchanged?
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_au; mkdir -p $O; : > $O/au.txt
K=10000000
timeout 900 python -m pytest tests/test_gpu_wire_pool.py tests/test_gpu_wire_dev.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2 | tee -a $O/au.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a $O/au.txt
mkdir -p /tmp/lablib; cp gubernator_amd/libguber_hip_lab.so /tmp/lablib/libguber_hip.so
for S in 8 1; do for T in 1 2 16; do for v in "GUBER_WIRE_DIRECT=1" "GUBER_WIRE_DIRECT=0"; do
  r=$(env $v LD_LIBRARY_PATH=/tmp/lablib timeout 120 tools/bench_pool_c $T $S 1 $K 1.0 200 wire 2>&1 | grep "^pool:" | sed -n 's/.*keys: *\([0-9.]*\) M decisions.*p50 \([0-9.]*\) us p99 \([0-9.]*\) us, conservation: [0-9]* keys [0-9]* decisions \([0-9]*\) violations.*/\1 M\/s p50 \2 p99 \3 violations \4/p')
  echo "[$v] $S table(s), $T caller(s) x 1-item RPCs: $r" | tee -a $O/au.txt
done; done; done
for v in "GUBER_WIRE_DIRECT=1" "GUBER_WIRE_DIRECT=0"; do
  r=$(env $v LD_LIBRARY_PATH=/tmp/lablib timeout 120 tools/bench_pool_c 192 8 1000 $K 2.0 200 wire 2>&1 | grep "^pool:" | sed -n 's/.*keys: *\([0-9.]*\) M decisions.*p50 \([0-9.]*\) us p99 \([0-9.]*\) us, conservation: [0-9]* keys [0-9]* decisions \([0-9]*\) violations.*/\1 M\/s p50 \2 p99 \3 violations \4/p')
  echo "[$v] 8 tables, 192 callers x 1000-item RPCs: $r" | tee -a $O/au.txt
done
