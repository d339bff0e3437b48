#!/bin/bash
# round 4, last call: the owner count following the traffic on the device (128 / 256 owners per batch) — the GPU suite, the headline
# with the uniform-key, leaky, expiring and one-table legs, the same with the count pinned either way, a kernel trace and the
# FETCH_SIZE / WRITE_SIZE passes of the headline; every command under its own short timeout
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=$R/gpurun_out/r04_final3; mkdir -p $O
timeout 120 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|Error" $O/pytest_gpu.txt | cut -c1-240 | head -8
timeout 90 python bench.py --no-cpu-baseline --extras "uniform,leaky,expiring,shards_1" > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
GUBER_PT_BITS=8 timeout 60 python bench.py --no-cpu-baseline --extras "uniform" --latency-steps 0 > $O/bench_pinned8.json 2> $O/bench_pinned8.err; echo "bench pinned 8 rc=$?"
GUBER_PT_BITS=7 timeout 60 python bench.py --no-cpu-baseline --extras "uniform" --latency-steps 0 > $O/bench_pinned7.json 2> $O/bench_pinned7.err; echo "bench pinned 7 rc=$?"
python - <<PY
import json
for f in ("bench", "bench_pinned8", "bench_pinned7"):
    try: d = json.load(open("$O/%s.json" % f))
    except Exception as e: print(f, "unreadable", e); continue
    print(f, "value", round(d["value"]/1e9, 3), "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "parity", str(d.get("parity"))[:60],
          {k: round(d[k]["value"]/1e9, 3) for k in ("uniform", "leaky", "expiring", "shards_1") if k in d and d[k].get("value")},
          "lat", d.get("batch_latency", {}).get("under_load", {}).get("p50"), d.get("batch_latency", {}).get("under_load", {}).get("p99"),
          {k: v for k, v in d["roofline"].get("kernel_avg_us", {}).items() if "multi" in k})
PY
cd /tmp && export TMPDIR=/tmp
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_fused -o t -- python $R/bench.py --no-cpu-baseline --extras= > $O/trace_fused.log 2>&1; echo "trace rc=$?"
f=$(find $O/trace_fused -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r04_final3_trace_fused_kernel_stats.csv && grep -E "k_part|k_own|k_eval3|k_front|k_eval2" $f | cut -c1-160 | head -8
rm -rf $O/trace_fused
