#!/bin/bash
# round 5, GPU call c: is the headline bound by the host's dispatcher?  (its thread burns CPU for the whole timed region: r05_b)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=$R/gpurun_out/r05_c; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu.txt | cut -c1-300
show() { python -c "import json,sys; d=json.load(open('$1')); t=d['timed_region']; print('$2', round(d['value']/1e9,3), 'G/s', d['ms_per_step']*1e3, 'us/step; enqueue wall', t['host_enqueue_ms'], 'busy', t['host_enqueue_busy_ms'], 'region', t['ms'])"; grep "\[dispatch\]" ${1%.json}.err | tail -2; }
X="--no-cpu-baseline --extras= --profile-steps 0 --latency-steps 0"
GUBER_DISPATCH_PROFILE=1 timeout 120 python bench.py $X > $O/b_head.json 2> $O/b_head.err; show $O/b_head.json headline
GUBER_DISPATCH_PROFILE=1 GUBER_FUSE_EP=0 timeout 120 python bench.py $X > $O/b_head_ep0.json 2> $O/b_head_ep0.err; show $O/b_head_ep0.json headline_ep0
# the same dispatcher, the GPU's share of the work shrunk: batches of 1024 / 4096 / 16384 requests over fewer keys
for nb in 1024 4096 16384; do
  GUBER_DISPATCH_PROFILE=1 timeout 120 python bench.py $X --batch $nb --keys 1000000 > $O/b_$nb.json 2> $O/b_$nb.err; show $O/b_$nb.json batch_$nb
done
