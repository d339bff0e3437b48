#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r02_z4; mkdir -p $O
digest='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(round(d["value"]/1e9,3), "G/s", d["ms_per_step"], "roofline:", r["kernel"], r.get("requests_per_launch"), r["achieved"], r["frac"], r["kernel_avg_us"], d["timed_region"]["repeats_of_the_step_list"])'
for cfg in "--dispatch one --shards 12 --streams 3" "--dispatch one --shards 16 --streams 4" "--dispatch one --shards 20 --streams 5" "--dispatch one --shards 12 --streams 4" "--dispatch threads --shards 4" "--dispatch threads --shards 12"; do
  echo "== default steps, $cfg" | tee -a $O/ab.txt
  timeout 400 python bench.py $cfg --extras "" --no-cpu-baseline 2>$O/err.txt | python -c "$digest" | tee -a $O/ab.txt
done
for cfg in "--dispatch one --shards 12 --streams 3 --algo leaky" "--dispatch one --shards 12 --streams 3 --dist uniform"; do
  echo "== driver cmd, $cfg" | tee -a $O/ab.txt
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 $cfg --extras "" --no-cpu-baseline 2>$O/err.txt | python -c "$digest" | tee -a $O/ab.txt
done
