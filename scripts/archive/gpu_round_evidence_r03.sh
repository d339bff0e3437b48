#!/bin/bash
# Runs on the GPU box: round 3's evidence in one call (everything except the rocprofv3 passes: scripts/gpu_profile_r03.sh).
#   GPU tests; the driver's bench command and the default bench line; GLOBAL on the native exchange (logical ranks);
#   phase stamps of k_front / k_eval2 (measurement build); dependent random-load latency by footprint; the pool surface by
#   caller threads and shards; batch-of-one latency through the C ABI.          usage: gpu_round_evidence_r03.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03_final}
cd $R
O=gpurun_out/$TAG; mkdir -p $O
(nproc; cat /sys/fs/cgroup/cpu.max; lscpu | grep -E "Model name|Socket|Core|Thread") > $O/host_cpus.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.txt
T0=$SECONDS
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench(driver cmd) rc=$? wall $((SECONDS-T0)) s"
T0=$SECONDS
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? wall $((SECONDS-T0)) s"
timeout 400 python bench.py --global-sync 8 --steps 64 --warmup 8 > $O/bench_global_sync.json 2> $O/bench_global_sync.err; echo "bench --global-sync rc=$?"
timeout 300 python bench.py --no-cpu-baseline --extras "" --router plain > $O/bench_plain_router.json 2>/dev/null; echo "bench(plain worker rule) rc=$?"
python - <<PY
import json
for f in ("bench_driver_cmd", "bench"):
    d=json.load(open("$O/%s.json" % f))
    print(f, "value", round(d["value"]/1e9,3), "ms/step", d["ms_per_step"], "lat", d["batch_latency"]["p50"], d["batch_latency"]["p99"], d["roofline"]["kernel_avg_us"], "frac", d["roofline"]["frac"], d["parity"][:40])
    for k in ("leaky","expiring","shards_1","uniform","end_to_end"):
        e=d.get(k,{}); print("   ", k, round((e.get("value") or 0)/1e9,3), e.get("ms_per_step"), (e.get("parity") or "")[:30], e.get("error"))
    print("    pool", {k: (v.get("value"), v.get("rpc_latency_us")) for k, v in d.get("pool", {}).items() if isinstance(v, dict)})
    print("    cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["by_threads"])
d=json.load(open("$O/bench_global_sync.json")); print("global-sync", round(d["value"]/1e9,3), d["global_sync"])
d=json.load(open("$O/bench_plain_router.json")); print("plain worker rule", round(d["value"]/1e9,3))
PY
export GUBER_HIP_LIB=$R/gubernator_amd/libguber_hip_timing.so
: > $O/phase_timing.txt
for a in "" "--dist uniform" "--algo leaky"; do
  echo "== timing build: bench.py --shards 1 $a" >> $O/phase_timing.txt
  timeout 300 python bench.py --no-cpu-baseline --shards 1 --steps 64 --min-batches 64 --profile-steps 0 --latency-steps 0 --extras "" $a 2>&1 | grep -A9 "phase timing" | grep -v '^{"metric' >> $O/phase_timing.txt
done
unset GUBER_HIP_LIB
cat $O/phase_timing.txt
timeout 300 tools/tlb_latency 2>&1 | grep -v burst | tee $O/dependent_load_latency.txt
{
for cfg in "1 8 1" "4 8 1" "16 8 1" "64 8 1" "16 1 1" "16 8 1000" "32 8 1000" "64 8 1000" "128 8 1000" "256 8 1000" "64 12 1000" "256 12 1000" "64 4 1000" "64 1 1000" "256 1 1000" "64 8 1000 cpp" "64 8 100"; do
  set -- $cfg
  timeout 120 tools/bench_pool_c $1 $2 $3 10000000 2.0 200 $4 2>&1 | grep -v amdgpu.ids
done
echo "== GUBER_POOL_ROUTED=0: stages per shard, callers sort by shard (what the front stage replaced)"
for cfg in "64 8 1000" "256 8 1000" "64 12 1000" "16 8 1"; do
  set -- $cfg
  GUBER_POOL_ROUTED=0 timeout 120 tools/bench_pool_c $1 $2 $3 10000000 2.0 200 2>&1 | grep -v amdgpu.ids
done
} | tee $O/pool_throughput.txt
tools/bench_config1_c | tee $O/config1.txt
