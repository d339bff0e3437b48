#!/bin/bash
# round 6, aj (an experiment that was taken out again; the script needs a build of that moment as libguber_hip.so and the commit before as libguber_hip_v_prev.so):
# against the library of the commit before (libguber_hip_v_prev.so), alternating on one box; the GPU tests of the front and the wire files first
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_aj; mkdir -p $O; : > $O/aj.txt
timeout 1200 python -m pytest tests/test_gpu_front.py tests/test_gpu_wire_pool.py tests/test_gpu_wire_dev.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2 | tee $O/tests.txt
K=10000000
mkdir -p /tmp/prev; cp gubernator_amd/libguber_hip_v_prev.so /tmp/prev/libguber_hip.so
for rep in 1 2 3; do for T in 1 64 128 192 256; do for v in new prev; do
  if [ $v = new ]; then L=""; else L=/tmp/prev; fi
  items=1000; [ $T = 1 ] && items=1
  r=$(LD_LIBRARY_PATH=$L timeout 120 tools/bench_pool_c $T 8 $items $K 1.5 200 wire 2>&1 | grep "^pool:" | sed -n 's/.*keys: *\([0-9.]*\) M decisions.*p50 \([0-9.]*\) us p99 \([0-9.]*\) us, conservation: [0-9]* keys [0-9]* decisions \([0-9]*\) violations.*/\1 M\/s p50 \2 p99 \3 violations \4/p')
  echo "rep $rep callers $T x $items $v: $r" | tee -a $O/aj.txt
done; done; done
