#!/bin/bash
# one shard on one device: callers that reserve first and touch every request once (GUBER_POOL_ONE_PASS, default 1; taken while
# everything outstanding fits two stages) against the general path, 32 .. 256 callers
cd /root/repo
mkdir -p gpurun_out
run() { timeout 120 tools/bench_pool_c $1 $2 $3 10000000 2.0 200 $4 2>&1 | grep -v amdgpu.ids | cut -c1-330; }
{
for op in 1 0; do echo "== GUBER_POOL_ONE_PASS=$op"; export GUBER_POOL_ONE_PASS=$op; run 32 1 1000; run 64 1 1000; run 128 1 1000; run 256 1 1000; done
unset GUBER_POOL_ONE_PASS
echo "== defaults, 8 shards"; run 64 8 1000; run 256 8 1000
} 2>&1 | tee gpurun_out/r03_onepass.txt
