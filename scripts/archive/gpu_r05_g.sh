#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=$R/gpurun_out/r05_g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "back_to_back or invalid_algorithm or duplicates_inside or eval3_and_the_next" > $O/pytest_new.txt 2>&1; echo "pytest new rc=$?"; tail -3 $O/pytest_new.txt | cut -c1-200
X="--no-cpu-baseline --extras= --shards 1 --min-batches 1024 --steps 1024 --profile-steps 256 --latency-steps 256"
for rep in 1 2; do for f in 0 1; do
  GUBER_ONE_TABLE_FUSED=$f timeout 120 python bench.py $X > $O/s1_f${f}_$rep.json 2> $O/s1_f${f}_$rep.err
  python -c "import json; d=json.load(open('$O/s1_f${f}_$rep.json')); print('one table, fused=$f', round(d['value']/1e9,3), d['ms_per_step'], d['roofline'].get('kernel_avg_us'), d['batch_latency']['idle']['p50'])"
done; done
