#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; O=gpurun_out/r04_v
GUBER_ENGINE_STATS=1 timeout 600 python bench.py --no-cpu-baseline --extras "" > ${O}_bench.json 2> ${O}_bench.err; echo "bench rc=$?"
grep "engine 0x" ${O}_bench.err | head -14 | cut -c1-400
python - <<PY
import json
for f in ("bench",):
    d = json.load(open("${O}_%s.json" % f)); print(f, "value", round(d["value"]/1e9, 3), "ms/step", d["ms_per_step"], "host_enqueue_ms", d["timed_region"].get("host_enqueue_ms"), "ms", d["timed_region"].get("ms"))
PY
