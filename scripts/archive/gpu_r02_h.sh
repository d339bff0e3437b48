#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r02_h
python tools/bench_global_native.py 2 100000 8
python tools/bench_global_native.py 8 100000 6
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_h/trace -o g -- python $R/tools/bench_global_native.py 2 100000 8 > /dev/null 2>&1
cd $R; f=$(find gpurun_out/r02_h/trace -name "*kernel_stats.csv" | head -1); head -25 $f | cut -c1-160
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "global" 2>&1 | tail -3
