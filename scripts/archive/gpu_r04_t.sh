#!/bin/bash
# round 4, step t: exact LRU — the tests that changed, the pool over binding caches, the whole GPU suite, the default bench line + one table
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; O=gpurun_out/r04_t
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_layer.py -m gpu -q -k "evicted_keys or larger_than_the_cache or routed_batches or compaction_keeps or caches_bind" > ${O}_pytest_lru.txt 2>&1; echo "lru rc=$?"; tail -15 ${O}_pytest_lru.txt | cut -c1-400
timeout 1500 python -m pytest tests -m gpu -q > ${O}_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|FAILED\|Error" ${O}_pytest_gpu.txt | cut -c1-300 | head -30
timeout 600 python bench.py --no-cpu-baseline --extras "" > ${O}_bench.json 2> ${O}_bench.err; echo "bench rc=$?"
timeout 600 python bench.py --no-cpu-baseline --extras "" --shards 1 > ${O}_bench_s1.json 2> ${O}_bench_s1.err; echo "bench s1 rc=$?"
python - <<PY
import json
for f in ("bench", "bench_s1"):
    d = json.load(open("${O}_%s.json" % f)); print(f, "value", round(d["value"]/1e9, 3), "ms/step", d["ms_per_step"], "roofline", d["roofline"].get("frac"), "parity", str(d.get("parity"))[:200])
PY
