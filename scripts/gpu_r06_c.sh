#!/bin/bash
# round 6: the `routed` leg at full size (10 M keys, 1024 timed batches) next to the pre-split headline, generations of 8 and 7 batches
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for gb in 8 7; do
  timeout 900 python bench.py --steps 1024 --min-batches 1024 --extra-batches 1024 --extras routed --gen-batches $gb --profile-steps 512 --latency-steps 64 --cpu-threads 32 > gpurun_out/r06_c_bench_gb$gb.json 2> gpurun_out/r06_c_bench_gb$gb.err
  tail -c 600 gpurun_out/r06_c_bench_gb$gb.err
  python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/r06_c_bench_gb$gb.json").read().strip().splitlines()[-1])
    r = j.get("routed", {})
    print("gb $gb: presplit", j["value"] / 1e9, "routed", (r.get("value") or 0) / 1e9, (r.get("parity") or r.get("error") or "")[:60])
    print("  kernels", r.get("kernel_avg_us"), r.get("front"))
    print("  latency", {k: (v or {}).get("p50") for k, v in (r.get("batch_latency") or {}).items()})
    print("  enqueue ms", r.get("host_enqueue_ms"), r.get("host_enqueue_busy_ms"), "timed", r.get("timed_ms"))
except Exception as ex:
    print("gb $gb: no line", ex)
PY
done
