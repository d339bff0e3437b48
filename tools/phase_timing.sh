#!/bin/bash
# Runs on the GPU box: per-workgroup phase timestamps of k_front / k_eval2 (measurement build, `make timing`).
# Output: gpurun_out/phase_timing.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
export GUBER_HIP_LIB=$R/gubernator_amd/libguber_hip_timing.so
: > gpurun_out/phase_timing.txt
for a in "" "--dist uniform" "--keys 200000" "--algo leaky"; do
  echo "== bench.py --shards 1 $a" >> gpurun_out/phase_timing.txt
  timeout 300 python bench.py --no-cpu-baseline --shards 1 --steps 64 --profile-steps 0 $a 2>&1 | grep -A9 "phase timing" | grep -v '^{"metric' >> gpurun_out/phase_timing.txt
done
cat gpurun_out/phase_timing.txt
