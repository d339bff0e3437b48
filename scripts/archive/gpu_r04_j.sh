#!/bin/bash
# is the headline bound by the host's enqueue?  The same dispatch with small batches (GPU work per launch shrinks, launches per step stay)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/${1:-r04_j}; mkdir -p $O
for b in 1024 4096 16384 65536; do
  for p in part claims; do
    GUBER_PIPELINE=$p timeout 600 python bench.py --no-cpu-baseline --extras "" --batch $b --keys 2000000 --profile-steps 0 --latency-steps 0 > $O/bench_${p}_$b.json 2> $O/bench_${p}_$b.err
    python - <<PY
import json
d=json.load(open("$O/bench_${p}_$b.json")); t=d["timed_region"]
print("$p batch $b: value", round(d["value"]/1e9,3), "G/s  us/step", round(d["ms_per_step"]*1e3,2), " enqueue ms", t["host_enqueue_ms"], "of", t["ms"])
PY
  done
done
