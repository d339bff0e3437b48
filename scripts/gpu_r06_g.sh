#!/bin/bash
# round 6: the default bench command as the driver runs it, the GPU suite, and the rocprofv3 evidence
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
bash scripts/gpu_tests.sh
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_g_bench_driver_cmd.json 2> gpurun_out/r06_g_bench_driver_cmd.err ) 2>&1 | tail -3
tail -c 800 gpurun_out/r06_g_bench_driver_cmd.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r06_g_bench_driver_cmd.json").read().strip().splitlines()[-1])
print("value", j["value"] / 1e9, "frac", j["roofline"]["frac"], "parity", j["parity"][:70])
print("measured", {k: v for k, v in (j["roofline"]["measured"] or {}).items() if k != "what"})
print("latency", {k: (v or {}).get("p50") for k, v in j["batch_latency"].items() if isinstance(v, dict)})
for k in ("presplit", "leaky", "expiring", "shards_1", "uniform", "end_to_end", "global_sync", "two_ranks"):
    e = j.get(k, {})
    print(k, e.get("value"), (e.get("parity") or e.get("error") or "")[:100])
for k, v in j.get("pool", {}).items():
    if isinstance(v, dict):
        print("pool", k, v.get("value"), (v.get("parity") or v.get("error") or "")[:50])
PY
bash scripts/gpu_profile_r06.sh r06 256 2>&1 | tail -60
