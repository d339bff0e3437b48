#!/bin/bash
# round 2, call B: A/B of the two k_front orderings (overlapped comparisons, 140 VGPRs vs lean, 105 VGPRs)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
Q="--no-cpu-baseline --extras '' --profile-steps 16"
for lib in hip hip_lean; do
  export GUBER_HIP_LIB=$R/gubernator_amd/libguber_$lib.so
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
  for s in 1 4 6; do
    echo "== $lib shards=$s"
    eval timeout 300 python bench.py $Q --shards $s 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k:d[k] for k in ('value','ms_per_step','batch_latency')}), d['roofline']['kernel_avg_us'])"
  done
done 2>&1 | tee gpurun_out/ab_b.txt
for lib in hip_timing hip_leantiming; do
  export GUBER_HIP_LIB=$R/gubernator_amd/libguber_$lib.so
  echo "== timing build $lib: bench.py --shards 1" >> gpurun_out/phase_timing_b.txt
  timeout 300 python bench.py --no-cpu-baseline --shards 1 --steps 64 --profile-steps 0 --extras "" 2>&1 | grep -A9 "phase timing" | grep -v '^{"metric' >> gpurun_out/phase_timing_b.txt
done
cat gpurun_out/phase_timing_b.txt
