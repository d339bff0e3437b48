#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r02_ab; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt | cut -c1-400
digest='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(round(d["value"]/1e9,3), "G/s", d["ms_per_step"], "roofline:", r["kernel"], r.get("requests_per_launch"), r["achieved"], r["frac"], r["kernel_avg_us"])'
for rep in 1 2; do
  echo "== driver cmd rep $rep" | tee -a $O/ab.txt
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --extras "" --no-cpu-baseline 2>$O/err.txt | python -c "$digest" | tee -a $O/ab.txt
done
echo "== default steps" | tee -a $O/ab.txt
timeout 400 python bench.py --extras "" --no-cpu-baseline 2>$O/err.txt | python -c "$digest" | tee -a $O/ab.txt
echo "== threads 4 shards, driver cmd" | tee -a $O/ab.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --dispatch threads --shards 4 --extras "" --no-cpu-baseline 2>$O/err.txt | python -c "$digest" | tee -a $O/ab.txt
