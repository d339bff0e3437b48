#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r02_v; mkdir -p $O
run() { echo "== end_to_end $*" | tee -a $O/e2e.txt; env "$@" timeout 400 python bench.py --no-cpu-baseline --extras end_to_end --profile-steps 0 --steps 32 --min-ms 30 2>$O/e2e.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['end_to_end']; d.pop('workload'); print(json.dumps(d))" | tee -a $O/e2e.txt; }
run GUBER_BENCH_E2E_DEPTH=2
run GUBER_BENCH_E2E_DEPTH=2 GUBER_STAGE_IN_WGS=256
run GUBER_BENCH_E2E_DEPTH=2 GUBER_STAGE_IN_WGS=512
run GUBER_BENCH_E2E_DEPTH=2 GUBER_STAGE_IN_WGS=128
run GUBER_BENCH_E2E_DEPTH=3
timeout 1200 python -m pytest tests -m gpu -q -x -k "stage or pool or host_layer or small" > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt | cut -c1-400
GUBER_STAGE_IN_WGS=256 timeout 1200 python -m pytest tests -m gpu -q -x -k "stage" > $O/pytest_gpu2.txt 2>&1; echo "pytest (copy kernel) rc=$?"; tail -3 $O/pytest_gpu2.txt | cut -c1-400
