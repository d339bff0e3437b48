#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r02_z2; mkdir -p $O
digest='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(round(d["value"]/1e9,3), "G/s", d["ms_per_step"], "roofline:", r["kernel"], r.get("requests_per_launch"), r["achieved"], r["frac"], r["kernel_avg_us"])'
for cfg in "--shards 4 --streams 1" "--shards 8 --streams 2" "--shards 4 --streams 2" "--shards 8 --streams 1" "--shards 12 --streams 3" "--shards 16 --streams 4"; do
  echo "== driver cmd, dispatch=one $cfg" | tee -a $O/ab.txt
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --dispatch one $cfg --extras "" --no-cpu-baseline 2>$O/err.txt | python -c "$digest" | tee -a $O/ab.txt
done
echo "== driver cmd, threads S=4" | tee -a $O/ab.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --extras "" --no-cpu-baseline 2>$O/err.txt | python -c "$digest" | tee -a $O/ab.txt
