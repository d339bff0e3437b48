#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r02_k
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r02_k/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r02_k/pytest_gpu.txt | cut -c1-400
timeout 400 python bench.py --no-cpu-baseline --extras end_to_end --profile-steps 0 --steps 32 --min-ms 30 2>gpurun_out/r02_k/e2e.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['end_to_end'], indent=1))" | tee gpurun_out/r02_k/e2e.txt; tail -3 gpurun_out/r02_k/e2e.err
