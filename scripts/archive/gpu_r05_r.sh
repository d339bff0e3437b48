#!/bin/bash
# occupancy variants on the final kernels, Zipf and uniform keys, alternating on one box:
#   e5   -DGUBER_EVAL2_WAVES=5                      k_eval2 / k_eval3 / k_evalpart at <= 96 VGPRs (five workgroups per CU; spills 228 B)
#   o4   -DGUBER_OWN_EPT=2 -DGUBER_OWN_WAVES=4      k_own at 128 VGPRs, 38 KB of LDS: four workgroups per CU (two messages per thread and round: rounds of 512)
#   o3e2 -DGUBER_OWN_EPT=2                          k_own with rounds of 512 at three workgroups per CU (what o4's round size alone does)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=$R/gpurun_out/r05_r; mkdir -p $O
X="--no-cpu-baseline --extras= --latency-steps 0 --profile-steps 256"
for rep in 1 2; do for v in default e5 o4 o3e2; do
  if [ $v = default ]; then unset GUBER_HIP_LIB; else export GUBER_HIP_LIB=$R/gubernator_amd/libguber_hip_v_$v.so; fi
  for dist in zipf uniform; do
    timeout 120 python bench.py $X --dist $dist > $O/${dist}_${v}_$rep.json 2> $O/${dist}_${v}_$rep.err
    python -c "import json; d=json.load(open('$O/${dist}_${v}_$rep.json')); print('$dist $v', round(d['value']/1e9,3), d['ms_per_step'], {k: v for k, v in d['roofline'].get('kernel_avg_us', {}).items() if 'own' in k or 'evalpart' in k})"
  done
done; done
unset GUBER_HIP_LIB
