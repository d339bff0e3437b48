#!/bin/bash
# headline as the driver runs it: one thread per shard vs one dispatcher with fused launches (roofline of the fused kernel)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r02_z; mkdir -p $O
digest='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(round(d["value"]/1e9,3), "G/s", d["ms_per_step"], "roofline:", r["kernel"], r.get("requests_per_launch"), r["achieved"], r["frac"], r["kernel_avg_us"], "traffic", r["traffic"], "pipeline", r["pipeline"]["frac"], d["parity"], d.get("pool",{}).get("value"))'
for rep in 1 2; do
for disp in threads one; do
  echo "== driver cmd, dispatch=$disp rep=$rep" | tee -a $O/ab.txt
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --dispatch $disp --extras pool 2>$O/err.txt | tee $O/last_$disp.json | python -c "$digest" | tee -a $O/ab.txt
  tail -2 $O/err.txt | cut -c1-300
done
done
echo "== default steps, dispatch=one" | tee -a $O/ab.txt
timeout 400 python bench.py --dispatch one --extras "" --no-cpu-baseline 2>$O/err.txt | python -c "$digest" | tee -a $O/ab.txt
