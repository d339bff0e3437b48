#!/bin/bash
# up to six tables per fused launch (GUBER_MULTI_MAX=6 build) vs four
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r02_z5; mkdir -p $O
digest='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(round(d["value"]/1e9,3), "G/s", d["ms_per_step"], "roofline:", r["kernel"], r.get("requests_per_launch"), r["achieved"], r["frac"], r["kernel_avg_us"])'
export GUBER_HIP_LIB=$R/gubernator_amd/libguber_hip_m6.so
for cfg in "--shards 12 --streams 2" "--shards 18 --streams 3" "--shards 12 --streams 3" "--shards 24 --streams 4"; do
  echo "== MULTI_MAX=6, driver cmd, $cfg" | tee -a $O/ab.txt
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --dispatch one $cfg --extras "" --no-cpu-baseline 2>$O/err.txt | python -c "$digest" | tee -a $O/ab.txt
done
echo "== MULTI_MAX=6, default steps, --shards 18 --streams 3" | tee -a $O/ab.txt
timeout 400 python bench.py --dispatch one --shards 18 --streams 3 --extras "" --no-cpu-baseline 2>$O/err.txt | python -c "$digest" | tee -a $O/ab.txt
unset GUBER_HIP_LIB
echo "== MULTI_MAX=4, driver cmd, --shards 12 --streams 3" | tee -a $O/ab.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --extras "" --no-cpu-baseline 2>$O/err.txt | python -c "$digest" | tee -a $O/ab.txt
