#!/bin/bash
# The pool with ONE front stage per device (guber_stage_submit_routed: the GPU hands the requests to the shards) against the
# arrangement it replaces (GUBER_POOL_ROUTED=0: stages per shard, callers sort by shard): parity first, then the pool surface.
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 | cut -c1-300 | tee gpurun_out/r03_routed_pytest.txt
run() { GUBER_POOL_DEBUG=1 timeout 120 tools/bench_pool_c $1 $2 $3 10000000 2.0 200 c 2>&1 | grep -v amdgpu.ids | cut -c1-420; }
for r in ${ROUNDS:-1 0 1}; do
  echo "== GUBER_POOL_ROUTED=$r"
  export GUBER_POOL_ROUTED=$r
  run 64 8 1000; run 32 8 1000; run 64 12 1000; run 256 8 1000
done 2>&1 | tee gpurun_out/r03_routed_ab.txt
