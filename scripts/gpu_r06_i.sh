#!/bin/bash
# round 6: generation size x depth of the front: rate and a generation's way through the GPU under load
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/r06_i_front_depth_gen.txt; : > $O
for gb in 8 16 24; do for d in 3 4 6 8; do
  GUBER_BENCH_FRONT_DEPTH=$d timeout 600 python bench.py --no-cpu-baseline --extras= --min-batches 1008 --steps 1008 --profile-steps 480 --latency-steps 48 --gen-batches $gb 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=j['batch_latency']
print('gen-batches $gb depth $d:', round(j['value']/1e9,3), 'G/s; generation under load p50/p99 us', l['under_load']['p50'], l['under_load']['p99'], 'idle p50', l['idle']['p50'], 'front', j['timed_region']['front'])" | tee -a $O
done; done
