#!/bin/bash
# round 6, ao: one table (the two-launch pipeline with claims) against the table's size — 10 M keys in 2^25 .. 2^28 slots (load 0.30 .. 0.04):
# fewer displaced keys = fewer second trips in k_front's chain?  (288 GB of HBM: a sparser table is affordable.)  And the routed headline at 4 x the slots.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
O=gpurun_out/r06_ao_table_slots.txt; : > $O
ARGS="--no-cpu-baseline --extras= --min-batches 1024 --steps 1024 --profile-steps 512 --latency-steps 64"
for rep in 1 2; do for lg in 0 25 26 27 28; do
  out=$(GUBER_BENCH_TABLE_SLOTS_LOG2=$lg timeout 600 python bench.py $ARGS --headline presplit --shards 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']/1e9,3), 'G/s; idle p50', d['batch_latency']['idle']['p50'], 'us; kernels', d['roofline'].get('kernel_avg_us'))")
  echo "rep $rep one table, slots 2^$lg (0 = the engine's rule): $out" | tee -a $O
done; done
for rep in 1 2; do for lg in 0 24 25; do
  out=$(GUBER_BENCH_TABLE_SLOTS_LOG2=$lg timeout 600 python bench.py $ARGS --headline routed 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']/1e9,3), 'G/s')")
  echo "rep $rep routed, 12 tables of 2^$lg slots each (0 = the engine's rule): $out" | tee -a $O
done; done
