#!/bin/bash
# round 4, step w: the 32-byte forms of the owner-partitioned pipeline (GUBER_PART_COMPACT) — the GPU suite on that build, then the
# headline on one box with: the default build, both halves, each half alone, and the measurement-only build without table accesses;
# the GPU suite on the default build; the pool with and without device routing; what random bucket accesses sustain
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; O=gpurun_out/r04_w; mkdir -p $O
L=$R/gubernator_amd
GUBER_HIP_LIB=$L/libguber_hip_v_compact.so timeout 500 python -m pytest tests -m gpu -q > $O/pytest_gpu_compact.txt 2>&1; echo "pytest (compact build) rc=$?"; grep -n "passed\|failed\|FAILED\|Error" $O/pytest_gpu_compact.txt | cut -c1-240 | head -12
run() {  # name, lib
  T0=$SECONDS
  GUBER_HIP_LIB=$L/$2 timeout 300 python bench.py --no-cpu-baseline --extras "" --latency-steps 0 > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 rc=$? wall $((SECONDS-T0)) s"
}
run base libguber_hip.so
run compact libguber_hip_v_compact.so
run msg32 libguber_hip_v_msg32.so
run rec32 libguber_hip_v_rec32.so
run compact_notable libguber_hip_v_compact_notable.so
run base2 libguber_hip.so
python - <<PY
import json
for f in ("base", "compact", "msg32", "rec32", "compact_notable", "base2"):
    try: d = json.load(open("$O/bench_%s.json" % f))
    except Exception as e: print(f, "unreadable", e); continue
    print(f, "value", round(d["value"]/1e9, 3), "ms/step", d["ms_per_step"], "kernels", d["roofline"].get("kernel_avg_us"))
PY
timeout 500 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.txt 2>&1; echo "pytest (default build) rc=$?"; grep -n "passed\|failed\|FAILED\|Error" $O/pytest_gpu.txt | cut -c1-240 | head -12
{
for dr in 0 1; do
  for cfg in "64 8 1000" "256 8 1000"; do
    set -- $cfg
    echo "GUBER_POOL_DEVROUTE=$dr"
    GUBER_POOL_DEVROUTE=$dr timeout 100 tools/bench_pool_c $1 $2 $3 10000000 2.0 200 2>&1 | grep -v amdgpu.ids
  done
done
} > $O/pool_devroute.txt 2>&1; cut -c1-300 $O/pool_devroute.txt
timeout 200 tools/random_access 2>&1 | grep -v amdgpu.ids | tee $O/random_access.txt | cut -c1-200
