#!/bin/bash
# round 6, ap: the payload stage's responses encoded on the device (k_wire_enc: one memcpy per RPC left for the caller) against the callers writing the
# varints themselves (GUBER_WIRE_HOST_ENCODE=1, laboratory build), alternating on one box; the wire pool tests first
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_ap${GUBER_AP_TAG:-}; mkdir -p $O; : > $O/ap.txt
export TMPDIR=/tmp
K=10000000
timeout 900 python -m pytest tests/test_gpu_wire_pool.py tests/test_gpu_wire_dev.py -m gpu -x -q 2>&1 | tail -3 | tee -a $O/ap.txt
mkdir -p /tmp/lablib; cp gubernator_amd/libguber_hip_lab.so /tmp/lablib/libguber_hip.so
for rep in 1 2; do for T in 64 128 192 256 384; do for v in "GUBER_WIRE_HOST_ENCODE=0" "GUBER_WIRE_HOST_ENCODE=1"; do
    r=$(env $v LD_LIBRARY_PATH=/tmp/lablib timeout 120 tools/bench_pool_c $T 8 1000 $K 2.0 200 wire 2>&1 | grep "^pool:" | sed -n 's/.*keys: *\([0-9.]*\) M decisions.*p50 \([0-9.]*\) us p99 \([0-9.]*\) us, conservation: [0-9]* keys [0-9]* decisions \([0-9]*\) violations.*/\1 M\/s p50 \2 p99 \3 violations \4/p')
    echo "rep $rep [$v] 8 tables, $T callers: $r" | tee -a $O/ap.txt
done; done; done
for T in 192 256; do for v in "GUBER_WIRE_HOST_ENCODE=0" "GUBER_WIRE_HOST_ENCODE=1"; do
    r=$(env $v LD_LIBRARY_PATH=/tmp/lablib timeout 120 tools/bench_pool_c $T 1 1000 $K 2.0 200 wire 2>&1 | grep "^pool:" | sed -n 's/.*keys: *\([0-9.]*\) M decisions.*p50 \([0-9.]*\) us p99 \([0-9.]*\) us, conservation: [0-9]* keys [0-9]* decisions \([0-9]*\) violations.*/\1 M\/s p50 \2 p99 \3 violations \4/p')
    echo "[$v] one table, $T callers: $r" | tee -a $O/ap.txt
    r=$(env $v LD_LIBRARY_PATH=/tmp/lablib timeout 120 tools/bench_pool_c $T 12 1000 $K 2.0 200 wire 2>&1 | grep "^pool:" | sed -n 's/.*keys: *\([0-9.]*\) M decisions.*p50 \([0-9.]*\) us p99 \([0-9.]*\) us, conservation: [0-9]* keys [0-9]* decisions \([0-9]*\) violations.*/\1 M\/s p50 \2 p99 \3 violations \4/p')
    echo "[$v] 12 tables, $T callers: $r" | tee -a $O/ap.txt
done; done
env LD_LIBRARY_PATH=/tmp/lablib timeout 120 tools/bench_pool_c 192 8 1000 $K 2.0 200 wire 2>&1 | tail -4 | tee -a $O/ap.txt
