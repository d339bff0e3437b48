#!/bin/bash
# the GLOBAL leg with two processes through the RCCL call sequence (tests/test_gpu_bench_multi.py): how often does its convergence check fail?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=$R/gpurun_out/r05_u; mkdir -p $O
bad=0
for i in $(seq 1 ${1:-12}); do
  timeout 300 python -m pytest tests/test_gpu_bench_multi.py -m gpu -q -x -k "global_leg" > $O/run_$i.txt 2>&1 || { bad=$((bad+1)); echo "run $i FAILED"; grep -E "global leg\]|did not converge" $O/run_$i.txt | head -5; }
done
echo "global leg with two processes: ${1:-12} runs, $bad failed"
