#!/bin/bash
# tables per fused launch x streams, owner-partitioned pipeline (same box): 4 x 3 (default), 6 x 2, 3 x 4, 6 x 3 (groups of 4), 12 shards
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/${1:-r04_i}; mkdir -p $O
run() {   # name, lib, streams, extra
  GUBER_HIP_LIB=$2 timeout 600 python bench.py --no-cpu-baseline --extras "" --streams $3 $4 > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 rc=$?"
}
run m4s3 $R/gubernator_amd/libguber_hip.so 3
run m6s2 $R/gubernator_amd/libguber_hip_vm6.so 2
run m4s4 $R/gubernator_amd/libguber_hip.so 4
run m6s3_18 $R/gubernator_amd/libguber_hip_vm6.so 3 "--shards 18"
run m4s2_8 $R/gubernator_amd/libguber_hip.so 2 "--shards 8"
python - <<PY
import json
for f in ("m4s3", "m6s2", "m4s4", "m6s3_18", "m4s2_8"):
    try: d = json.load(open("$O/bench_%s.json" % f))
    except Exception as e: print(f, "unreadable", e); continue
    print(f, "value", round(d["value"]/1e9, 3), "ms/step", d["ms_per_step"], "kernels", d.get("roofline", {}).get("kernel_avg_us"), "under load", {k: d["batch_latency"]["under_load"][k] for k in ("p50","p99")})
PY
