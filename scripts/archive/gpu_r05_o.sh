#!/bin/bash
# uniform keys run 256 owners per batch: four tables per launch are 1 024 k_own workgroups for 768 places (3 per CU).  Shards x streams
# (= tables per launch) for uniform and for Zipf keys, on one box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=$R/gpurun_out/r05_o; mkdir -p $O
X="--no-cpu-baseline --extras= --latency-steps 0 --profile-steps 256"
for rep in 1 2; do
for cfg in "12 3" "12 4" "12 6" "9 3" "15 5" "18 6"; do
  set -- $cfg
  for dist in uniform zipf; do
    timeout 120 python bench.py $X --dist $dist --shards $1 --streams $2 > $O/${dist}_$1x$2_$rep.json 2> $O/${dist}_$1x$2_$rep.err
    python -c "import json; d=json.load(open('$O/${dist}_$1x$2_$rep.json')); print('$dist shards $1 streams $2:', round(d['value']/1e9,3), d['ms_per_step'], {k: v for k, v in d['roofline'].get('kernel_avg_us', {}).items() if 'own' in k or 'evalpart' in k})"
  done
done; done
