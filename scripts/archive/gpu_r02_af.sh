#!/bin/bash
# counter read-backs riding on k_front: full GPU tests + the two bench lines (the round's last numbers)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r02_final6; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt | cut -c1-300
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench(driver cmd) rc=$?"
timeout 600 python bench.py --extras "" --cpu-threads 32 --cpu-seconds 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
for f in ("bench_driver_cmd", "bench"):
    d=json.load(open("$O/%s.json" % f))
    print(f, "value", round(d["value"]/1e9,3), "ms/step", d["ms_per_step"], d["roofline"]["kernel_avg_us"], "frac", d["roofline"]["frac"], d["parity"])
    for k in ("leaky","shards_1","uniform","end_to_end","pool"):
        e=d.get(k,{}); 
        if e: print("   ", k, round(e.get("value",0)/1e9,3), e.get("ms_per_step"), e.get("parity"), e.get("error"))
PY
