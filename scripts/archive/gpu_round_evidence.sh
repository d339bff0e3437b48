#!/bin/bash
# Runs on the GPU box: the round's evidence in one call.  usage: gpu_round_evidence.sh <tag>
#   GPU tests; the driver's bench command and the default bench line; phase timestamps (measurement build); rocprofv3 kernel
#   trace + PMC passes (scripts/gpu_profile_r02.sh); batch-of-one latency through the C ABI; the native GLOBAL exchange.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02_final}
cd $R
O=gpurun_out/$TAG; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.txt | cut -c1-200
T0=$SECONDS
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench(driver cmd) rc=$? wall $((SECONDS-T0)) s"
T0=$SECONDS
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? wall $((SECONDS-T0)) s"
timeout 300 python bench.py --no-cpu-baseline --extras "" --dispatch threads --shards 4 > $O/bench_threads_4_shards.json 2>/dev/null; echo "bench(thread per shard, 4 shards) rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_threads_4_shards.json")); print("one thread + stream per shard, 4 shards:", round(d["value"]/1e9,3), "G/s", d["ms_per_step"])
for f in ("bench_driver_cmd", "bench"):
    d=json.load(open("$O/%s.json" % f))
    print(f, "value", round(d["value"]/1e9,3), "ms/step", d["ms_per_step"], "timed ms", d["timed_region"]["ms"], "lat", d["batch_latency"]["p50"], d["batch_latency"]["p99"], d["roofline"]["kernel_avg_us"], "frac", d["roofline"]["frac"], d["parity"])
    for k in ("leaky","shards_1","uniform","end_to_end"):
        e=d.get(k,{}); print("   ", k, round(e.get("value",0)/1e9,3), e.get("ms_per_step"), e.get("batch_latency",{}).get("p50"), e.get("parity"), e.get("error"))
    print("    cpu", d["cpu_baseline"]["by_threads"])
PY
export GUBER_HIP_LIB=$R/gubernator_amd/libguber_hip_timing.so
: > $O/phase_timing.txt
for a in "" "--dist uniform" "--algo leaky"; do
  echo "== timing build: bench.py --shards 1 $a" >> $O/phase_timing.txt
  timeout 300 python bench.py --no-cpu-baseline --shards 1 --steps 64 --profile-steps 0 --extras "" $a 2>&1 | grep -A9 "phase timing" | grep -v '^{"metric' >> $O/phase_timing.txt
done
unset GUBER_HIP_LIB
cat $O/phase_timing.txt
./scripts/gpu_profile_r02.sh $TAG 2>&1 | tail -32
tools/bench_config1_c | tee $O/config1.txt
python tools/bench_global_native.py 2 100000 8 | tee $O/global_native.txt
python tools/bench_global_native.py 4 100000 6 | tee -a $O/global_native.txt
