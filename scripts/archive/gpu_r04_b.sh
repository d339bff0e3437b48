#!/bin/bash
# phase stamps of k_part / k_own / k_eval3 (measurement build), one table, Zipf and uniform keys
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r04_b; mkdir -p $O
export GUBER_HIP_LIB=$R/gubernator_amd/libguber_hip_timing.so
: > $O/phase_timing.txt
for a in "" "--dist uniform"; do
  echo "== timing build: bench.py --shards 1 $a" >> $O/phase_timing.txt
  timeout 300 python bench.py --no-cpu-baseline --shards 1 --steps 64 --min-batches 64 --profile-steps 0 --latency-steps 0 --extras "" $a 2>&1 | grep -A9 "phase timing" | grep -v '^{"metric' >> $O/phase_timing.txt
done
cat $O/phase_timing.txt
