#!/bin/bash
# the driver's form of the bench command passes --warmup 5: five batches do not reach every one of the 12 shards before the clock starts.
# Does the timed region carry first-use costs then?  --warmup 5 / 16 / 48, alternating on one box (same flags otherwise)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=$R/gpurun_out/r05_t; mkdir -p $O
X="--gpus 1 --steps 20 --no-cpu-baseline --extras= --latency-steps 0 --profile-steps 256"
for rep in 1 2 3; do for w in 5 16 48; do
  timeout 120 python bench.py $X --warmup $w > $O/w${w}_$rep.json 2> $O/w${w}_$rep.err
  python -c "import json; d=json.load(open('$O/w${w}_$rep.json')); t=d['timed_region']; print('warmup $w:', round(d['value']/1e9,3), d['ms_per_step'], 'timed ms', t['ms'], 'enqueue', t['host_enqueue_ms'], [s['stream_ms'] for s in t['shard_streams']][:3])"
done; done
