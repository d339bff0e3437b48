#!/bin/bash
# round 4, step u: the bench lines with CacheSize = 2 x the resident keys (12 shards, one table)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; O=gpurun_out/r04_u
timeout 600 python bench.py --no-cpu-baseline --extras "" > ${O}_bench.json 2> ${O}_bench.err; echo "bench rc=$?"
timeout 600 python bench.py --no-cpu-baseline --extras "" --shards 1 > ${O}_bench_s1.json 2> ${O}_bench_s1.err; echo "bench s1 rc=$?"
python - <<PY
import json
for f in ("bench", "bench_s1"):
    d = json.load(open("${O}_%s.json" % f)); print(f, "value", round(d["value"]/1e9, 3), "ms/step", d["ms_per_step"], "roofline", d["roofline"].get("frac"), "parity", str(d.get("parity"))[:200], "host_enqueue_ms", d["timed_region"].get("host_enqueue_ms"), "ms", d["timed_region"].get("ms"))
PY
