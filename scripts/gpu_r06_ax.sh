#!/bin/bash
# round 6, ax: the direct path for RPCs of up to FOUR requests: the wire and host-layer files of the GPU suite, smoke(), then 1 / 2 / 4 / 5-item RPCs from 1 and 4 callers
# (5 items: through the stages), 8 tables and one; GUBER_WIRE_DIRECT=0 for comparison (laboratory build)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_ax; mkdir -p $O; : > $O/ax.txt
K=10000000
timeout 900 python -m pytest tests/test_gpu_wire_pool.py tests/test_gpu_wire_dev.py tests/test_gpu_host_layer.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2 | tee -a $O/ax.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a $O/ax.txt
mkdir -p /tmp/lablib; cp gubernator_amd/libguber_hip_lab.so /tmp/lablib/libguber_hip.so
for S in 8 1; do for T in 1 4; do for items in 1 2 4 5; do for v in "GUBER_WIRE_DIRECT=8" "GUBER_WIRE_DIRECT=0"; do
  r=$(env $v LD_LIBRARY_PATH=/tmp/lablib timeout 120 tools/bench_pool_c $T $S $items $K 0.7 150 wire 2>&1 | grep "^pool:" | sed -n 's/.*keys: *\([0-9.]*\) M decisions.*p50 \([0-9.]*\) us p99 \([0-9.]*\) us, conservation: [0-9]* keys [0-9]* decisions \([0-9]*\) violations.*/\1 M\/s p50 \2 p99 \3 violations \4/p')
  echo "[$v] $S table(s), $T caller(s) x $items-item RPCs: $r" | tee -a $O/ax.txt
done; done; done; done
