#!/bin/bash
# k_own v3 (first message requested before the scan, two LDS atomics per message, 384 keys / 768 messages per round):
# GPU suite, bench with the default policy (fused launches: owner-partitioned; single launches: claims), part everywhere, stamps
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/${1:-r04_e}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt | cut -c1-300
run() {   # name, pipeline env
  GUBER_PIPELINE=$2 timeout 600 python bench.py --no-cpu-baseline --extras shards_1,uniform,leaky > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 rc=$?"
}
run auto ""
run part part
run claims claims
python - <<PY
import json
for f in ("auto", "part", "claims"):
    try: d = json.load(open("$O/bench_%s.json" % f))
    except Exception as e: print(f, "unreadable", e); continue
    print(f, "value", round(d["value"]/1e9, 3), "ms/step", d["ms_per_step"], "lat", d.get("batch_latency", {}).get("p50"), "kernels", d.get("roofline", {}).get("kernel_avg_us"))
    for k in ("shards_1", "uniform", "leaky"):
        e = d.get(k) or {}
        print("   ", k, round((e.get("value") or 0)/1e9, 3), e.get("ms_per_step"), (e.get("batch_latency") or {}).get("p50"), e.get("error"))
PY
export GUBER_HIP_LIB=$R/gubernator_amd/libguber_hip_timing.so
for a in "" "--dist uniform"; do
  echo "== timing build: GUBER_PIPELINE=part bench.py --shards 1 $a"
  GUBER_PIPELINE=part timeout 300 python bench.py --no-cpu-baseline --shards 1 --steps 64 --min-batches 64 --profile-steps 0 --latency-steps 0 --extras "" $a 2>&1 | grep -A9 "phase timing" | grep -v '^{"metric'
done | tee $O/phase_timing.txt
