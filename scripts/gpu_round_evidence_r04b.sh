#!/bin/bash
# Runs on the GPU box: the round's closing evidence on the final code (32-byte records, k_part at six workgroups per CU, the bounded LDS
# insert loop) in ONE call: the GPU suite, the bench line as the driver runs it, rocprofv3 kernel traces (12 shards fused; one table with
# either pipeline) and the FETCH_SIZE / WRITE_SIZE passes, summarised and checked against the line by tools/summarize_r04.py.
#   usage: gpu_round_evidence_r04b.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04_final2}
cd $R
O=gpurun_out/$TAG; mkdir -p $O
(nproc; cat /sys/fs/cgroup/cpu.max; lscpu | grep -E "Model name|Socket|Core|Thread") > $O/host_cpus.txt 2>&1
timeout 150 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|device wire decode|native global sync" $O/pytest_gpu.txt | cut -c1-300
T0=$SECONDS
timeout 220 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench(driver cmd) rc=$? wall $((SECONDS-T0)) s"
python - <<PY
import json
d=json.load(open("$O/bench_driver_cmd.json"))
print("driver cmd: value", round(d["value"]/1e9,3), "ms/step", d["ms_per_step"], "roofline", {k: d["roofline"][k] for k in ("achieved","frac","traffic")}, d["roofline"]["kernel_avg_us"])
print("   latency idle", d["batch_latency"]["idle"]["p50"], d["batch_latency"]["idle"]["p99"], "under load", {k: d["batch_latency"]["under_load"][k] for k in ("p50","p99","n")})
print("   parity", d["parity"][:160])
for k in ("leaky","expiring","shards_1","uniform","end_to_end"):
    e=d.get(k,{}); print("   ", k, round((e.get("value") or 0)/1e9,3), e.get("ms_per_step"), (e.get("parity") or "")[:70], e.get("error"))
print("    pool", {k: (v.get("value"), v.get("rpc_latency_us")) for k, v in d.get("pool", {}).items() if isinstance(v, dict)})
print("    cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
PMC_SETS="FETCH_SIZE WRITE_SIZE" PMC_SETS_S1="FETCH_SIZE WRITE_SIZE" bash scripts/gpu_profile_r04.sh ${TAG}_prof 256 2>&1 | grep -v "^|" | tail -25
