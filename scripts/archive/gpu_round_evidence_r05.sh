#!/bin/bash
# Runs on the GPU box: the round's evidence in one call (everything except the rocprofv3 passes: scripts/gpu_profile_r05.sh).
#   GPU tests; the driver's bench command; GLOBAL on the native exchange (logical ranks); the pool surface; batch-of-one latency.
#   usage: gpu_round_evidence_r05.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05_final}
cd $R
O=gpurun_out/$TAG; mkdir -p $O
(nproc; cat /sys/fs/cgroup/cpu.max; lscpu | grep -E "Model name|Socket|Core|Thread") > $O/host_cpus.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|LRU divergence|device wire decode|native global sync" $O/pytest_gpu.txt | cut -c1-300
T0=$SECONDS
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench(driver cmd) rc=$? wall $((SECONDS-T0)) s"
timeout 400 python bench.py --global-sync 8 --steps 64 --warmup 8 > $O/bench_global_sync.json 2> $O/bench_global_sync.err; echo "bench --global-sync rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_driver_cmd.json"))
print("driver cmd: value", round(d["value"]/1e9,3), "ms/step", d["ms_per_step"], "roofline", {k: d["roofline"][k] for k in ("achieved","frac","traffic")}, d["roofline"]["kernel_avg_us"])
print("   latency idle", d["batch_latency"]["idle"]["p50"], d["batch_latency"]["idle"]["p99"], "under load", {k: d["batch_latency"]["under_load"][k] for k in ("p50","p99","n")})
print("   issue", d["roofline"].get("issue"), "enqueue wall/busy", d["timed_region"]["host_enqueue_ms"], d["timed_region"]["host_enqueue_busy_ms"], "steps", d["steps"], d["steps_requested"])
print("   parity", d["parity"][:160])
for k in ("leaky","expiring","shards_1","uniform","end_to_end"):
    e=d.get(k,{}); print("   ", k, round((e.get("value") or 0)/1e9,3), e.get("ms_per_step"), (e.get("parity") or "")[:70], e.get("error"))
print("    pool", {k: (v.get("value"), v.get("rpc_latency_us")) for k, v in d.get("pool", {}).items() if isinstance(v, dict)})
print("    cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
g=json.load(open("$O/bench_global_sync.json")); print("global-sync", round(g["value"]/1e9,3), g["global_sync"])
PY
{
for cfg in "64 8 1000" "256 8 1000" "64 12 1000" "64 1 1000" "16 8 1"; do
  set -- $cfg
  timeout 120 tools/bench_pool_c $1 $2 $3 10000000 2.0 200 2>&1 | grep -v amdgpu.ids
done
} | tee $O/pool_throughput.txt
tools/bench_config1_c | tee $O/config1.txt
