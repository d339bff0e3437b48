#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r02_u; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x -k "stage or pool or host_layer or small" > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt | cut -c1-400
for depth in 2 3; do
  echo "== end_to_end depth=$depth (copy kernel in, results in place)" | tee -a $O/e2e.txt
  GUBER_BENCH_E2E_DEPTH=$depth timeout 400 python bench.py --no-cpu-baseline --extras end_to_end --profile-steps 0 --steps 32 --min-ms 30 2>$O/e2e.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['end_to_end']; d.pop('workload'); print(json.dumps(d))" | tee -a $O/e2e.txt; tail -2 $O/e2e.err | cut -c1-300
done
echo "== end_to_end depth=2 GUBER_STAGE_OUT_DMA=1" | tee -a $O/e2e.txt
GUBER_STAGE_OUT_DMA=1 GUBER_BENCH_E2E_DEPTH=2 timeout 400 python bench.py --no-cpu-baseline --extras end_to_end --profile-steps 0 --steps 32 --min-ms 30 2>$O/e2e.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['end_to_end']; d.pop('workload'); print(json.dumps(d))" | tee -a $O/e2e.txt
