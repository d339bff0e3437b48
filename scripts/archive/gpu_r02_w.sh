#!/bin/bash
# lean mode of k_front (table lines only for likely claimers): parity + rate A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r02_w; mkdir -p $O
GUBER_LEAN=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "not full_size and not epoch_wraps" > $O/pytest_lean.txt 2>&1; echo "pytest(lean) rc=$?"; tail -3 $O/pytest_lean.txt | cut -c1-400
digest='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]/1e9,3), "G/s", d["ms_per_step"], d["roofline"]["kernel_avg_us"])'
for rep in 1 2; do
for lean in 0 1; do
  if [ $lean = 1 ]; then export GUBER_LEAN=1; else unset GUBER_LEAN; fi
  for cfg in "--shards 1" "--shards 4" "--shards 4 --dispatch one" "--shards 8" "--shards 4 --dist uniform"; do
    echo "== lean=$lean $cfg" | tee -a $O/ab.txt
    timeout 300 python bench.py --no-cpu-baseline --extras "" $cfg 2>$O/err.txt | python -c "$digest" | tee -a $O/ab.txt
  done
done
done
