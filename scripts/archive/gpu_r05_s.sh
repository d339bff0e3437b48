#!/bin/bash
# k_own with rounds of 1 024 messages (-DGUBER_OWN_EPT=4: four messages per thread; 168 VGPRs, 40 B of scratch, 44 KB of LDS) against the
# default's 768: fewer rounds that split for a hot owner's long list.  Zipf and uniform keys, alternating on one box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=$R/gpurun_out/r05_s; mkdir -p $O
X="--no-cpu-baseline --extras= --latency-steps 0 --profile-steps 256"
for rep in 1 2 3; do for v in default e4; do
  if [ $v = default ]; then unset GUBER_HIP_LIB; else export GUBER_HIP_LIB=$R/gubernator_amd/libguber_hip_v_$v.so; fi
  for dist in zipf uniform; do
    timeout 120 python bench.py $X --dist $dist > $O/${dist}_${v}_$rep.json 2> $O/${dist}_${v}_$rep.err
    python -c "import json; d=json.load(open('$O/${dist}_${v}_$rep.json')); print('$dist $v', round(d['value']/1e9,3), d['ms_per_step'], {k: v for k, v in d['roofline'].get('kernel_avg_us', {}).items() if 'own' in k or 'evalpart' in k})"
  done
done; done
unset GUBER_HIP_LIB
