#!/bin/bash
# the bench line as the driver runs it, with the round-4 bench (fresh request columns per batch, all timed batches checked by digest,
# latency under load, roofline.frac = pipeline)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/${1:-r04_h}; mkdir -p $O
T0=$SECONDS
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench(driver cmd) rc=$? wall $((SECONDS-T0)) s"
tail -5 $O/bench_driver_cmd.err | cut -c1-400
python - <<PY
import json
d=json.load(open("$O/bench_driver_cmd.json"))
print("value", round(d["value"]/1e9,3), "ms/step", d["ms_per_step"], "roofline", {k: d["roofline"][k] for k in ("achieved","frac","kernel","traffic")}, d["roofline"]["kernel_avg_us"])
print("latency", d["batch_latency"]["idle"]["p50"], d["batch_latency"]["idle"]["p99"], "under load", d["batch_latency"]["under_load"])
print("parity", d["parity"][:200])
for k in ("leaky","expiring","shards_1","uniform","end_to_end"):
    e=d.get(k,{}); print("   ", k, round((e.get("value") or 0)/1e9,3), e.get("ms_per_step"), (e.get("parity") or "")[:60], e.get("error"), e.get("roofline_frac"), (e.get("batch_latency") or {}).get("under_load"))
print("    pool", {k: (v.get("value"), v.get("rpc_latency_us")) for k, v in d.get("pool", {}).items() if isinstance(v, dict)})
print("    cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["by_threads"])
PY
