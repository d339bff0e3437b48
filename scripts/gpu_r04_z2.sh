#!/bin/bash
# round 4, step z2/z3 (z3: k_own in passes; prev = the build before): after the bound on k_own's LDS insert loop — the regression test on the GPU (default build and 128 owners per batch),
# then 128 owners against 256 on the headline and the uniform-key control; every command under its own short timeout
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; O=gpurun_out/r04_z3; mkdir -p $O
L=$R/gubernator_amd
timeout 90 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "more_keys_of_one_owner" > $O/pytest_hang_default.txt 2>&1; echo "regression test (default build) rc=$?"; tail -2 $O/pytest_hang_default.txt | cut -c1-200
GUBER_HIP_LIB=$L/libguber_hip_v_p7.so timeout 90 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "more_keys_of_one_owner" > $O/pytest_hang_p7.txt 2>&1; echo "regression test (p7 build) rc=$?"; tail -2 $O/pytest_hang_p7.txt | cut -c1-200
run() {  # name, lib
  GUBER_HIP_LIB=$L/$2 timeout 60 python bench.py --no-cpu-baseline --extras "uniform" --latency-steps 0 > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 rc=$?"
}
NAMES="prev base p7 prev_again base_again p7_again"
run prev libguber_hip_v_prev.so
run base libguber_hip.so
run p7 libguber_hip_v_p7.so
run prev_again libguber_hip_v_prev.so
run base_again libguber_hip.so
run p7_again libguber_hip_v_p7.so
python - <<PY
import json
for f in "$NAMES".split():
    try: d = json.load(open("$O/bench_%s.json" % f))
    except Exception as e: print(f, "unreadable", e); continue
    print(f, "value", round(d["value"]/1e9, 3), "ms/step", d["ms_per_step"], "uniform", round(d.get("uniform", {}).get("value", 0)/1e9, 3), "kernels", {k: v for k, v in d["roofline"].get("kernel_avg_us", {}).items() if "multi" in k})
PY
