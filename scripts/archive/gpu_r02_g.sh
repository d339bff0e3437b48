#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r02_g
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_g/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/r02_g/pytest_gpu.txt | cut -c1-400
for s in 4 8 1; do
  echo -n "== shards=$s  "
  timeout 300 python bench.py --no-cpu-baseline --extras= --profile-steps 16 --shards $s 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,3),'G/s', d['ms_per_step']*1e3,'us/step p50', d['batch_latency']['p50'], d['roofline']['kernel_avg_us'])"
done
