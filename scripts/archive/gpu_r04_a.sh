#!/bin/bash
# Round 4, first GPU pass: the GPU suite (new owner-partitioned pipeline behind flag 64 and by default for device-resident batches),
# then the bench A/B: two-launch pipeline with claims (GUBER_PIPELINE=claims) against the owner-partitioned one, same box.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r04_a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.txt | cut -c1-300
for mode in claims part; do
  GUBER_PIPELINE=$mode timeout 600 python bench.py --no-cpu-baseline --extras shards_1,uniform > $O/bench_$mode.json 2> $O/bench_$mode.err; echo "bench $mode rc=$?"
done
timeout 600 python bench.py --extras leaky > $O/bench_part_parity.json 2> $O/bench_part_parity.err; echo "bench part+parity rc=$?"
python - <<PY
import json
for f in ("bench_claims", "bench_part", "bench_part_parity"):
    try:
        d = json.load(open("$O/%s.json" % f))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, "value", round(d["value"]/1e9, 3), "ms/step", d["ms_per_step"], "lat", d.get("batch_latency", {}).get("p50"), (d.get("parity") or "")[:60])
    print("   kernels", d.get("roofline", {}).get("kernel_avg_us"), d.get("roofline", {}).get("requests_per_launch"))
    for k in ("leaky", "shards_1", "uniform"):
        e = d.get(k) or {}
        print("   ", k, round((e.get("value") or 0)/1e9, 3), e.get("ms_per_step"), (e.get("batch_latency") or {}).get("p50"), (e.get("roofline") or {}).get("kernel_avg_us"), (e.get("parity") or "")[:30], e.get("error"))
PY
tail -5 $O/*.err | cut -c1-300
