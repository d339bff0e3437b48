#!/bin/bash
# stages with DMA copies: full GPU tests, then end-to-end A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r02_q; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.txt | cut -c1-400
for dma in 1 0; do
  if [ $dma = 0 ]; then export GUBER_NO_STAGE_DMA=1; else unset GUBER_NO_STAGE_DMA; fi
  echo "== end_to_end dma=$dma" | tee -a $O/e2e.txt
  timeout 400 python bench.py --no-cpu-baseline --extras end_to_end --profile-steps 0 --steps 32 --min-ms 30 2>$O/e2e.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(d['end_to_end'], indent=1))" | tee -a $O/e2e.txt; tail -3 $O/e2e.err | cut -c1-300
done
