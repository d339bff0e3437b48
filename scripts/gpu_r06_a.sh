#!/bin/bash
# round 6, first GPU call: the device-resident front (guber_front_*) — parity on the GPU, then a short `routed` leg beside the pre-split headline
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_front.py -x -q 2>&1 | tail -15 > gpurun_out/r06_a_front_tests.txt
cat gpurun_out/r06_a_front_tests.txt
for gb in 8 4 16 1; do
  timeout 600 python bench.py --steps 512 --min-batches 512 --extra-batches 512 --extras routed --gen-batches $gb --profile-steps 256 --latency-steps 64 --cpu-threads 32 > gpurun_out/r06_a_bench_gb$gb.json 2> gpurun_out/r06_a_bench_gb$gb.err
  tail -c 600 gpurun_out/r06_a_bench_gb$gb.err
  python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/r06_a_bench_gb$gb.json").read().strip().splitlines()[-1])
    r = j.get("routed", {})
    print("gb $gb: presplit", j["value"] / 1e9, "routed", (r.get("value") or 0) / 1e9, r.get("parity", r.get("error"))[:80] if r else None)
    print("  kernels", r.get("kernel_avg_us"), r.get("front"), r.get("batch_latency"))
    print("  enqueue ms", r.get("host_enqueue_ms"), r.get("host_enqueue_busy_ms"), "timed", r.get("timed_ms"))
except Exception as ex:
    print("gb $gb: no line", ex)
PY
done
