#!/bin/bash
# round 4, step y: occupancy — k_part with the requests' fields in the bitmaps' LDS (27 KB, 73 VGPRs: six workgroups per CU instead of
# four), k_eval3 at five waves per SIMD, k_own at four, with and without 32-byte records; one box, the headline configuration
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; O=gpurun_out/r04_y; mkdir -p $O
L=$R/gubernator_amd
run() {  # name, lib
  GUBER_HIP_LIB=$L/$2 timeout 300 python bench.py --no-cpu-baseline --extras "" --latency-steps 0 > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 rc=$?"
}
NAMES="base lds ev5 own4 all5 all5r p5r base_again lds_again"
run base libguber_hip.so
run lds libguber_hip_v_lds.so
run ev5 libguber_hip_v_ev5.so
run own4 libguber_hip_v_own4.so
run all5 libguber_hip_v_all5.so
run all5r libguber_hip_v_all5r.so
run p5r libguber_hip_v_p5r.so
run base_again libguber_hip.so
run lds_again libguber_hip_v_lds.so
python - <<PY
import json
for f in "$NAMES".split():
    try: d = json.load(open("$O/bench_%s.json" % f))
    except Exception as e: print(f, "unreadable", e); continue
    print(f, "value", round(d["value"]/1e9, 3), "ms/step", d["ms_per_step"], "kernels", {k: v for k, v in d["roofline"].get("kernel_avg_us", {}).items() if "multi" in k})
PY
GUBER_HIP_LIB=$L/libguber_hip_v_p5r.so timeout 500 python -m pytest tests -m gpu -q > $O/pytest_gpu_p5r.txt 2>&1; echo "pytest (p5r build) rc=$?"; grep -n "passed\|failed\|FAILED\|Error" $O/pytest_gpu_p5r.txt | cut -c1-240 | head -12
