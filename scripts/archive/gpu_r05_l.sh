#!/bin/bash
# the device wire decoder with a workgroup per 8 KB WINDOW (k_wire_win_a + k_wire_win_b), the batch numbered by k_wire_scan's last
# workgroup and k_wire_fill parsing from LDS: parity against the host transcoder, the rates by RPC size through the C call, and the
# kernel time per decode BY SHAPE from a kernel trace (the shape = k_wire_scan's grid: one workgroup per payload)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=$R/gpurun_out/r05_l; mkdir -p $O
for s in 1 0; do
  GUBER_WIRE_SERIAL=$s timeout 300 python -m pytest tests/test_gpu_wire_dev.py -m gpu -q -s > $O/pytest_wire_serial$s.txt 2>&1; echo "wire decode, GUBER_WIRE_SERIAL=$s rc=$?"
  grep -E "device wire decode|passed|failed" $O/pytest_wire_serial$s.txt | cut -c1-200
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/wire_trace -o t -- python -m pytest $R/tests/test_gpu_wire_dev.py -m gpu -q -k "report_throughput" > $O/wire_trace.log 2>&1; echo "trace rc=$?"
f=$(find $O/wire_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -i "wire" $f | cut -c1-200 && cp $f $O/r05_wire_decode_kernel_stats.csv
t=$(find $O/wire_trace -name "*kernel_trace.csv" | head -1)
python $R/tools/wire_trace_by_shape.py "$t" | tee $O/wire_by_shape.txt
rm -rf $O/wire_trace
