#!/bin/bash
# round 4, step o: table footprint / load factor vs throughput, the retry-reorder test
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; O=gpurun_out/r04_o
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "retries_may_run" > ${O}_retry_test.txt 2>&1; tail -30 ${O}_retry_test.txt | cut -c1-250
run() {  # name, slots_log2, extra
  GUBER_BENCH_TABLE_SLOTS_LOG2=$2 timeout 600 python bench.py --no-cpu-baseline --extras "" $3 > ${O}_bench_$1.json 2> ${O}_bench_$1.err; echo "bench $1 rc=$?"
}
run s12_2p23 23 ""
run s12_2p24 24 ""
run s1_2p26 26 "--shards 1"
python - <<PY
import json
for f in ("s12_2p23", "s12_2p24", "s1_2p26"):
    try: d = json.load(open("${O}_bench_%s.json" % f))
    except Exception as e: print(f, "unreadable", e); continue
    print(f, "value", round(d["value"]/1e9, 3), "ms/step", d["ms_per_step"])
PY
