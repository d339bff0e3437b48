#!/bin/bash
# round 4, the last seconds of GPU time: k_eval3 as two launches (GUBER_EVAL3_SPLIT=1) — one parity test with it, then the headline
# (with the expiring leg: the slow launch has work there) with and without it
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=$R/gpurun_out/r04_split; mkdir -p $O
GUBER_EVAL3_SPLIT=1 timeout 25 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "either_owner_count or adversarial_streams and 64" > $O/pytest_split.txt 2>&1; echo "pytest(split) rc=$?"; tail -1 $O/pytest_split.txt | cut -c1-160
GUBER_EVAL3_SPLIT=1 timeout 20 python bench.py --no-cpu-baseline --extras "expiring" --latency-steps 0 > $O/bench_split.json 2> $O/bench_split.err; echo "bench split rc=$?"
timeout 20 python bench.py --no-cpu-baseline --extras "expiring" --latency-steps 0 > $O/bench_base.json 2> $O/bench_base.err; echo "bench base rc=$?"
python - <<PY
import json
for f in ("bench_split", "bench_base"):
    try: d = json.load(open("$O/%s.json" % f))
    except Exception as e: print(f, "unreadable", e); continue
    print(f, "value", round(d["value"]/1e9, 3), "ms/step", d["ms_per_step"], "expiring", round(d.get("expiring", {}).get("value", 0)/1e9, 3), {k: v for k, v in d["roofline"].get("kernel_avg_us", {}).items() if "multi" in k})
PY
