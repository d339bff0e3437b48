#!/bin/bash
# round 4, step p: k_own with the displaced keys' directory entries requested together with the home bucket (-DGUBER_OWN_DIR_EARLY=1), default and larger tables
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; O=gpurun_out/r04_p
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "retries_follow" > ${O}_retry_test.txt 2>&1; tail -4 ${O}_retry_test.txt | cut -c1-250
run() {  # name, lib, slots_log2, extra
  GUBER_HIP_LIB=$2 GUBER_BENCH_TABLE_SLOTS_LOG2=$3 timeout 600 python bench.py --no-cpu-baseline --extras "" $4 > ${O}_bench_$1.json 2> ${O}_bench_$1.err; echo "bench $1 rc=$?"
}
D=$R/gubernator_amd/libguber_hip.so; V=$R/gubernator_amd/libguber_hip_v_direarly.so
run base_22 $D 0 ""
run early_22 $V 0 ""
run base_23 $D 23 ""
run early_23 $V 23 ""
run early_s1part $V 0 "--shards 1"
python - <<PY
import json
for f in ("base_22", "early_22", "base_23", "early_23", "early_s1part"):
    try: d = json.load(open("${O}_bench_%s.json" % f))
    except Exception as e: print(f, "unreadable", e); continue
    print(f, "value", round(d["value"]/1e9, 3), "ms/step", d["ms_per_step"], d.get("parity", {}).get("digest") if isinstance(d.get("parity"), dict) else "")
PY
