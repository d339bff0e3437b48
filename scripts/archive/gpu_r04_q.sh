#!/bin/bash
# round 4, step q: the GPU suite with the roomier default tables, the default bench line, one table
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; O=gpurun_out/r04_q
timeout 1500 python -m pytest tests -m gpu -q -x > ${O}_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" ${O}_pytest_gpu.txt | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline --extras "" > ${O}_bench.json 2> ${O}_bench.err; echo "bench rc=$?"
timeout 600 python bench.py --no-cpu-baseline --extras "" --shards 1 > ${O}_bench_s1.json 2> ${O}_bench_s1.err; echo "bench s1 rc=$?"
python - <<PY
import json
for f in ("bench", "bench_s1"):
    d = json.load(open("${O}_%s.json" % f)); print(f, "value", round(d["value"]/1e9, 3), "ms/step", d["ms_per_step"], "roofline", d["roofline"].get("frac"), "parity", str(d.get("parity"))[:200])
PY
