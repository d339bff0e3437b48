#!/bin/bash
# routed pool: admission limit (callers in the CPU part of a call) and depth around the defaults; then the bench line on the same library
cd /root/repo
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_host_layer.py -m gpu -q 2>&1 | tail -3
run() { timeout 120 tools/bench_pool_c $1 $2 $3 10000000 2.0 200 c 2>&1 | grep -v amdgpu.ids | cut -c1-330; }
{
for ma in 12 13 14 15; do echo "== GUBER_POOL_MAX_ACTIVE=$ma"; GUBER_POOL_MAX_ACTIVE=$ma run 64 8 1000; GUBER_POOL_MAX_ACTIVE=$ma run 256 8 1000; done
echo "== GUBER_POOL_EAGER_MIN=8192"; GUBER_POOL_EAGER_MIN=8192 run 64 8 1000
echo "== GUBER_POOL_EAGER_MIN=2048"; GUBER_POOL_EAGER_MIN=2048 run 64 8 1000
echo "== GUBER_POOL_NT_STORES=0"; GUBER_POOL_NT_STORES=0 run 64 8 1000
echo "== small RPCs"; run 16 8 1; run 64 8 100; run 64 1 1000
} 2>&1 | tee gpurun_out/r03_routed_knobs.txt
timeout 600 python bench.py --no-cpu-baseline --extras=shards_1,pool 2>/dev/null | tee gpurun_out/r03_routed_bench.json | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']/1e9,3), round(j['shards_1']['value']/1e9,3), json.dumps(j.get('pool'))[:900])"
