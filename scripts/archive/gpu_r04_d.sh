#!/bin/bash
# same-box A/B of the two-launch pipeline's k_front: round 3's (vr3) / displaced keys in one trip (vc2) / + table fetch with the claim
# only for groups of <= 2 (vs2) / <= 1 (default build); then the owner-partitioned pipeline on the default build; GPU suite first.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/${1:-r04_d}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt | cut -c1-300
run() {   # name, lib, pipeline
  GUBER_HIP_LIB=$2 GUBER_PIPELINE=$3 timeout 600 python bench.py --no-cpu-baseline --extras shards_1,uniform > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 rc=$?"
}
run vr3 $R/gubernator_amd/libguber_hip_vr3.so claims
run vc2 $R/gubernator_amd/libguber_hip_vc2.so claims
run vs2 $R/gubernator_amd/libguber_hip_vs2.so claims
run vs1 $R/gubernator_amd/libguber_hip.so claims
run part $R/gubernator_amd/libguber_hip.so part
python - <<PY
import json
for f in ("vr3", "vc2", "vs2", "vs1", "part"):
    try: d = json.load(open("$O/bench_%s.json" % f))
    except Exception as e: print(f, "unreadable", e); continue
    print(f, "value", round(d["value"]/1e9, 3), "ms/step", d["ms_per_step"], "lat", d.get("batch_latency", {}).get("p50"), "kernels", d.get("roofline", {}).get("kernel_avg_us"))
    for k in ("shards_1", "uniform"):
        e = d.get(k) or {}
        print("   ", k, round((e.get("value") or 0)/1e9, 3), e.get("ms_per_step"), (e.get("batch_latency") or {}).get("p50"), e.get("error"))
PY
