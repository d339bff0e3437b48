#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r02_i
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r02_i/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r02_i/pytest_gpu.txt | cut -c1-300
python tools/bench_config1.py 2>/dev/null | tee gpurun_out/r02_i/config1.txt
GUBER_NO_ZEROCOPY=1 python tools/bench_config1.py 2>/dev/null | sed 's/^/copy path: /' | tee -a gpurun_out/r02_i/config1.txt
for z in 0 1; do
  if [ $z = 1 ]; then export GUBER_NO_ZEROCOPY=1; else unset GUBER_NO_ZEROCOPY; fi
  echo "== end_to_end (GUBER_NO_ZEROCOPY=$z)"
  timeout 300 python bench.py --no-cpu-baseline --extras end_to_end --profile-steps 0 --steps 32 --min-ms 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['end_to_end'])"
done 2>&1 | tee gpurun_out/r02_i/e2e.txt
