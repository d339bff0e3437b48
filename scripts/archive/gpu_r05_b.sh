#!/bin/bash
# round 5, GPU call b: the GPU suite on the round's code (fused launch default, the N > 1 bench rehearsal), the headline with the
# dispatcher's CPU time beside its wall time, and the SQ counters of the three kernels before the instruction diet
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=$R/gpurun_out/r05_b; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.txt | cut -c1-300
for rep in 1 2; do
  timeout 120 python bench.py --no-cpu-baseline --extras "" > $O/bench_$rep.json 2> $O/bench_$rep.err; echo "bench rc=$?"
  python -c "import json; d=json.load(open('$O/bench_$rep.json')); t=d['timed_region']; print(round(d['value']/1e9,3), d['ms_per_step'], 'enqueue wall', t['host_enqueue_ms'], 'busy', t['host_enqueue_busy_ms'], 'region', t['ms'], d['batch_latency']['under_load'])"
done
bash scripts/gpu_r05_sq.sh r05_b_sq_ep0 GUBER_FUSE_EP=0 2>&1 | grep -v "^    " | tail -12
