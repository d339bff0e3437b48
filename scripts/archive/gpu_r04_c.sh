#!/bin/bash
# k_own v2 (34 KB LDS, 128 VGPRs, one message read, references in LDS): GPU suite, bench (12 shards / one table / uniform), phase stamps
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/${1:-r04_c}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline --extras shards_1,uniform > $O/bench_part.json 2> $O/bench_part.err; echo "bench part rc=$?"
python - <<PY
import json
for f in ("bench_part",):
    d = json.load(open("$O/%s.json" % f))
    print(f, "value", round(d["value"]/1e9, 3), "ms/step", d["ms_per_step"], "lat", d.get("batch_latency", {}).get("p50"))
    print("   kernels", d.get("roofline", {}).get("kernel_avg_us"), d.get("roofline", {}).get("requests_per_launch"))
    for k in ("shards_1", "uniform"):
        e = d.get(k) or {}
        print("   ", k, round((e.get("value") or 0)/1e9, 3), e.get("ms_per_step"), (e.get("batch_latency") or {}).get("p50"), (e.get("roofline") or {}).get("kernel_avg_us"), e.get("error"))
PY
export GUBER_HIP_LIB=$R/gubernator_amd/libguber_hip_timing.so
for a in ""; do
  echo "== timing build: bench.py --shards 1 $a"
  timeout 300 python bench.py --no-cpu-baseline --shards 1 --steps 64 --min-batches 64 --profile-steps 0 --latency-steps 0 --extras "" $a 2>&1 | grep -A9 "phase timing" | grep -v '^{"metric'
done | tee $O/phase_timing.txt
