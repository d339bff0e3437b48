#!/bin/bash
# round 2: GPU tests + headline bench + phase timing + rocprof/PMC summary.  usage: gpu_r02_e.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02_e}
cd $R
mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/$TAG/pytest_gpu.txt | cut -c1-300
timeout 600 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err; echo "bench rc=$?"; python - <<PY
import json
d=json.load(open("gpurun_out/$TAG/bench.json"))
print("value", d["value"]/1e9, "ms/step", d["ms_per_step"], "lat", d["batch_latency"]["p50"], d["batch_latency"]["p99"], d["roofline"]["kernel_avg_us"], d["parity"])
for k in ("leaky","shards_1","uniform","end_to_end"):
    e=d.get(k,{}); print(k, e.get("value",0)/1e9, e.get("ms_per_step"), e.get("batch_latency",{}).get("p50"), e.get("kernel_avg_us"), e.get("parity"), e.get("error"))
print("cpu", d["cpu_baseline"]["by_threads"])
PY
tail -3 gpurun_out/$TAG/bench.err
export GUBER_HIP_LIB=$R/gubernator_amd/libguber_hip_timing.so
echo "== timing build: bench.py --shards 1" > gpurun_out/$TAG/phase_timing.txt
timeout 300 python bench.py --no-cpu-baseline --shards 1 --steps 64 --profile-steps 0 --extras "" 2>&1 | grep -A9 "phase timing" | grep -v '^{"metric' >> gpurun_out/$TAG/phase_timing.txt
cat gpurun_out/$TAG/phase_timing.txt
unset GUBER_HIP_LIB
./scripts/gpu_profile_r02.sh $TAG 2>&1 | tail -30
