#!/bin/bash
# round 6: the front's timeline under rocprofv3 (who is busy, how many kernels at a time)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
for gb in 16 8; do
cd /tmp && rm -rf /tmp/prof_f && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_f -o fr -- python $R/tools/front_probe.py 10000000 $gb $((1024/gb)) > /tmp/prof_f.log 2>&1
grep "in one call" /tmp/prof_f.log
f=$(find /tmp/prof_f -name '*kernel_trace.csv' | head -1)
python $R/tools/front_timeline.py "$f" | tee $R/gpurun_out/r06_f_timeline_gb$gb.txt
done
