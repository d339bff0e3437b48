#!/bin/bash
# load-aware slot placement vs plain consistent hash of the keys over the logical shards
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r02_ac; mkdir -p $O
digest='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(round(d["value"]/1e9,3), "G/s", d["ms_per_step"], "roofline:", r["kernel"], r.get("requests_per_launch"), r["achieved"], r["frac"], r["kernel_avg_us"], d["parity"], d["config"]["placement"])'
run() { echo "== $*" | tee -a $O/ab.txt; timeout 400 python bench.py "$@" --extras "" 2>$O/err.txt | python -c "$digest" | tee -a $O/ab.txt; tail -1 $O/err.txt | cut -c1-200; }
run --gpus 1 --steps 20 --warmup 5 --router slots --cpu-threads 32 --cpu-seconds 1
run --gpus 1 --steps 20 --warmup 5 --router ring --no-cpu-baseline
run --router slots --no-cpu-baseline
run --router ring --no-cpu-baseline
run --gpus 1 --steps 20 --warmup 5 --router slots --shards 16 --streams 4 --no-cpu-baseline
run --gpus 1 --steps 20 --warmup 5 --router slots --shards 9 --streams 3 --no-cpu-baseline
run --gpus 1 --steps 20 --warmup 5 --router slots --shards 20 --streams 5 --no-cpu-baseline
run --gpus 1 --steps 20 --warmup 5 --router slots --shards 8 --streams 2 --no-cpu-baseline
run --gpus 1 --steps 20 --warmup 5 --router slots --algo leaky --no-cpu-baseline
