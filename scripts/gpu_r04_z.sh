#!/bin/bash
# round 4, step z: 128 owners per batch instead of 256 (fewer, fuller k_own workgroups), with 4 messages per thread and round, with
# room for 384 keys per round; the headline and the uniform-key control on one box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; O=gpurun_out/r04_z; mkdir -p $O
L=$R/gubernator_amd
run() {  # name, lib
  GUBER_HIP_LIB=$L/$2 timeout 300 python bench.py --no-cpu-baseline --extras "uniform" --latency-steps 0 > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 rc=$?"
}
NAMES="base p7 p7e4 p7k base_again p7_again"
run base libguber_hip.so
run p7 libguber_hip_v_p7.so
run p7e4 libguber_hip_v_p7e4.so
run p7k libguber_hip_v_p7k.so
run base_again libguber_hip.so
run p7_again libguber_hip_v_p7.so
python - <<PY
import json
for f in "$NAMES".split():
    try: d = json.load(open("$O/bench_%s.json" % f))
    except Exception as e: print(f, "unreadable", e); continue
    print(f, "value", round(d["value"]/1e9, 3), "ms/step", d["ms_per_step"], "uniform", round(d.get("uniform", {}).get("value", 0)/1e9, 3), "kernels", {k: v for k, v in d["roofline"].get("kernel_avg_us", {}).items() if "multi" in k})
PY
GUBER_HIP_LIB=$L/libguber_hip_v_p7.so timeout 500 python -m pytest tests -m gpu -q > $O/pytest_gpu_p7.txt 2>&1; echo "pytest (p7 build) rc=$?"; grep -n "passed\|failed\|FAILED\|Error" $O/pytest_gpu_p7.txt | cut -c1-240 | head -12
