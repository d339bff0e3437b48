#!/bin/bash
# host launch rate of the box and one-dispatcher vs thread-per-shard enqueueing
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r02_n; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/launch_rate tools/launch_rate.hip -lpthread 2>/dev/null && timeout 300 /tmp/launch_rate | tee $O/launch_rate.txt
digest='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]/1e9,3), "G/s", d["ms_per_step"], d["roofline"]["kernel_avg_us"], d.get("parity",{}).get("ok") if isinstance(d.get("parity"),dict) else d.get("parity")); print("   ", d["timed_region"]["shard_streams"])'
for disp in one threads; do
for S in 2 4 8; do
  echo "== zipf S=$S dispatch=$disp" | tee -a $O/dispatch.txt
  timeout 300 python bench.py --no-cpu-baseline --extras "" --shards $S --dispatch $disp 2>$O/err.txt | python -c "$digest" | tee -a $O/dispatch.txt
done
done
echo "== uniform S=4 dispatch=one" | tee -a $O/dispatch.txt
timeout 300 python bench.py --no-cpu-baseline --extras "" --shards 4 --dist uniform --dispatch one 2>$O/err.txt | python -c "$digest" | tee -a $O/dispatch.txt
