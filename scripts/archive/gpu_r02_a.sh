#!/bin/bash
# round 2, call A: GPU tests + headline bench + phase timing of the new kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.txt
timeout 600 python bench.py > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err; echo "bench rc=$?"; cat gpurun_out/bench_a.json; tail -5 gpurun_out/bench_a.err
export GUBER_HIP_LIB=$R/gubernator_amd/libguber_hip_timing.so
for a in "" "--algo leaky"; do
  echo "== timing build: bench.py --shards 1 $a" >> gpurun_out/phase_timing_a.txt
  timeout 300 python bench.py --no-cpu-baseline --shards 1 --steps 64 --profile-steps 0 --extras "" $a 2>&1 | grep -A9 "phase timing" | grep -v '^{"metric' >> gpurun_out/phase_timing_a.txt
done
cat gpurun_out/phase_timing_a.txt
