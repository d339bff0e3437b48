#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, bench lines, rocprofv3 kernel stats.
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
if [ "$1" != "notest" ]; then
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -15
fi
timeout 900 python bench.py > gpurun_out/bench_token.json 2> gpurun_out/bench_token.err; echo "rc=$?"; tail -3 gpurun_out/bench_token.err; cat gpurun_out/bench_token.json
timeout 600 python bench.py --algo leaky --steps 128 --no-cpu-baseline > gpurun_out/bench_leaky.json 2> gpurun_out/bench_leaky.err; echo "rc=$?"; cat gpurun_out/bench_leaky.json
timeout 600 python bench.py --dist uniform --steps 128 --no-cpu-baseline > gpurun_out/bench_uniform.json 2> gpurun_out/bench_uniform.err; echo "rc=$?"; cat gpurun_out/bench_uniform.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_token
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_token -o token -- python $R/bench.py --steps 128 --warmup 16 --no-cpu-baseline --profile-steps 0 > $R/gpurun_out/prof_token.log 2>&1; echo "rocprof rc=$?"
ls -R $R/gpurun_out/prof_token | head; head -20 $R/gpurun_out/prof_token/*kernel_stats.csv
