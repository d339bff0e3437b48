#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r02_j
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r02_j/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r02_j/pytest_gpu.txt | cut -c1-300
tools/bench_config1_c | tee gpurun_out/r02_j/config1.txt
python tools/bench_config1.py 2>/dev/null | tee -a gpurun_out/r02_j/config1.txt
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_j/trace -o c1 -- $R/tools/bench_config1_c > /dev/null 2>&1
cd $R; f=$(find gpurun_out/r02_j/trace -name "*kernel_stats.csv" | head -1); head -5 $f | cut -c1-200
