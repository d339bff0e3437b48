#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=$R/gpurun_out/r05_f; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_bench_multi.py -m gpu -q -x -s > $O/pytest_multi.txt 2>&1; echo "pytest multi rc=$?"; grep -E "passed|failed|ratio|G/s" $O/pytest_multi.txt | tail -5
T0=$SECONDS
timeout 600 python bench.py --no-cpu-baseline --extras "" > $O/bench.json 2> $O/bench.err; echo "bench rc=$? wall $((SECONDS-T0)) s"
python -c "import json; d=json.load(open('$O/bench.json')); print(round(d['value']/1e9,3), d['ms_per_step'], d['batch_latency']['under_load'], d['batch_latency']['idle'], d['roofline'].get('issue'))"
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_bench_multi.py > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt | cut -c1-200
