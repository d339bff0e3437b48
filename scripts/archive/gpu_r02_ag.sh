#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r02_final7; mkdir -p $O
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench(driver cmd) rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_driver_cmd.json"))
print("value", round(d["value"]/1e9,3), "ms/step", d["ms_per_step"], "timed ms", d["timed_region"]["ms"], "repeats", d["timed_region"]["repeats_of_the_step_list"], d["roofline"]["kernel"], d["roofline"]["frac"], d["parity"])
print(d["timed_region"]["enqueue"]); print(d["config"]["workload"][-200:])
for k in ("leaky","shards_1","uniform","end_to_end","pool"):
    e=d.get(k,{}); print("   ", k, round(e.get("value",0)/1e9,3), e.get("ms_per_step"), e.get("parity"), e.get("error"))
PY
