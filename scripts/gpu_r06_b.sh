#!/bin/bash
# round 6: where does the front's time go — the probe alone (generations one at a time: per-kernel times with nothing else in flight; a run of
# generations in one call), and under rocprofv3's kernel trace
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_front.py -x -q 2>&1 | tail -3
for gb in 8 4 16; do
timeout 300 python tools/front_probe.py 2000000 $gb 32 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_b_probe.txt
done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o fr -- python $R/tools/front_probe.py 2000000 8 32 > /tmp/prof_b.log 2>&1
tail -3 /tmp/prof_b.log
f=$(find /tmp/prof_b -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/r06_b_front_kernel_stats.csv
head -30 "$f" | cut -c1-200
