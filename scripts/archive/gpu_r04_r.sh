#!/bin/bash
# round 4, step r: exact LRU (eviction pre-pass) — the LRU tests first, then the whole GPU suite, then the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; O=gpurun_out/r04_r
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "evicted_keys or larger_than_the_cache or lrucache or live_set or cache_operations" > ${O}_pytest_lru.txt 2>&1; echo "lru rc=$?"; tail -5 ${O}_pytest_lru.txt | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q > ${O}_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|FAILED\|Error" ${O}_pytest_gpu.txt | cut -c1-300 | head -30
timeout 600 python bench.py --no-cpu-baseline --extras "" > ${O}_bench.json 2> ${O}_bench.err; echo "bench rc=$?"
python - <<PY
import json
for f in ("bench",):
    d = json.load(open("${O}_%s.json" % f)); print(f, "value", round(d["value"]/1e9, 3), "ms/step", d["ms_per_step"], "roofline", d["roofline"].get("frac"), "parity", str(d.get("parity"))[:200])
PY
