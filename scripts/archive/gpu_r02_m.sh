#!/bin/bash
# what bounds the multi-shard rate: transaction ceilings of the box, per-shard stream times, HW queue count
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r02_m; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fabric_peak tools/fabric_peak.hip 2>/dev/null && timeout 300 /tmp/fabric_peak | tee $O/fabric_peak.txt
digest='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]/1e9,3), "G/s", d["ms_per_step"], d["roofline"]["kernel_avg_us"]); print("   ", d["timed_region"]["shard_streams"])'
for S in 1 2 4 8 16; do
  echo "== zipf S=$S" | tee -a $O/shards.txt
  timeout 300 python bench.py --no-cpu-baseline --extras "" --shards $S 2>$O/err.txt | python -c "$digest" | tee -a $O/shards.txt
done
for S in 8 16; do
  echo "== zipf S=$S GPU_MAX_HW_QUEUES=8" | tee -a $O/shards.txt
  GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --no-cpu-baseline --extras "" --shards $S 2>$O/err.txt | python -c "$digest" | tee -a $O/shards.txt
done
for S in 1 4 8; do
  echo "== uniform S=$S" | tee -a $O/shards.txt
  timeout 300 python bench.py --no-cpu-baseline --extras "" --dist uniform --shards $S 2>$O/err.txt | python -c "$digest" | tee -a $O/shards.txt
done
