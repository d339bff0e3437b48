#!/bin/bash
# round 6: stability — the GPU suite three times, the driver's command twice
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2 3; do timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -1; done
for i in 1 2; do
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_l_bench_$i.json 2> gpurun_out/r06_l_bench_$i.err ) 2>&1 | grep real
python - <<PY
import json
j = json.loads(open("gpurun_out/r06_l_bench_$i.json").read().strip().splitlines()[-1])
print("value", round(j["value"] / 1e9, 3), "frac", j["roofline"]["frac"], "lat", j["batch_latency"]["under_load"]["p50"], j["batch_latency"]["under_load"]["p99"])
print({k: (round(j[k]["value"] / 1e9, 3) if j.get(k, {}).get("value") else j.get(k, {}).get("error", "no value")) for k in ("presplit", "routed_leaky", "leaky", "expiring", "shards_1", "uniform", "end_to_end", "global_sync", "two_ranks")})
print({k: round(v["value"] / 1e6, 1) for k, v in j["pool"].items() if isinstance(v, dict) and v.get("value")})
PY
done
