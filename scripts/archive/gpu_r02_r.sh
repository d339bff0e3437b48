#!/bin/bash
# stages: counter snapshots by kernel, 2 vs 3 batches in flight
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r02_r; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x -k "stage or pool or host_layer or small" > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt | cut -c1-400
for depth in 2 3 4; do
  echo "== end_to_end depth=$depth" | tee -a $O/e2e.txt
  GUBER_BENCH_E2E_DEPTH=$depth timeout 400 python bench.py --no-cpu-baseline --extras end_to_end --profile-steps 0 --steps 32 --min-ms 30 2>$O/e2e.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['end_to_end']; d.pop('workload'); print(json.dumps(d))" | tee -a $O/e2e.txt; tail -2 $O/e2e.err | cut -c1-300
done
