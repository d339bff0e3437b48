#!/bin/bash
# the device wire decoder: the parallel chain walk (k_wire_scan_par + the serial walk for what it leaves) against the serial walk
# alone — parity against the host transcoder on both, the rates by RPC size, and a kernel trace of the parallel form
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=$R/gpurun_out/r05_h; mkdir -p $O
for s in 1 0; do
  GUBER_WIRE_SERIAL=$s timeout 300 python -m pytest tests/test_gpu_wire_dev.py -m gpu -q -s > $O/pytest_wire_serial$s.txt 2>&1; echo "wire decode, GUBER_WIRE_SERIAL=$s rc=$?"
  grep -E "device wire decode|passed|failed" $O/pytest_wire_serial$s.txt | cut -c1-200
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/wire_trace -o t -- python -m pytest $R/tests/test_gpu_wire_dev.py -m gpu -q -k "report_throughput" > $O/wire_trace.log 2>&1; echo "trace rc=$?"
f=$(find $O/wire_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -i "wire" $f | cut -c1-200 && cp $f $O/r05_wire_decode_kernel_stats.csv
rm -rf $O/wire_trace
