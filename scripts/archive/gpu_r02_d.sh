#!/bin/bash
# round 2, call D: ablation of k_front / k_eval2 phases (timing only — ablated builds give wrong answers)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
Q="--no-cpu-baseline --extras= --profile-steps 16 --steps 128"
for lib in hip hip_abl1 hip_abl2 hip_abl4 hip_abl8 hip_abl48 hip_abl64 hip_abl127; do
  export GUBER_HIP_LIB=$R/gubernator_amd/libguber_$lib.so
  for s in 4 1; do
    echo -n "== $lib shards=$s  "
    timeout 300 python bench.py $Q --shards $s 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,3),'G/s', d['ms_per_step']*1e3,'us/step p50', d['batch_latency']['p50'], d['roofline']['kernel_avg_us'])"
  done
done 2>&1 | tee gpurun_out/ablate_d.txt
