#!/bin/bash
# k_own: the second key of a round of more than 256 keys (uniform keys) touched beside the first (-DGUBER_OWN_PREFETCH2=1) against the
# default build, alternating on one box: uniform keys, and the headline (which must not lose)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=$R/gpurun_out/r05_n; mkdir -p $O
GUBER_HIP_LIB=$R/gubernator_amd/libguber_hip_v_pf2.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "uniform or owner or part or routed" > $O/pytest_parity.txt 2>&1; echo "pytest parity (variant) rc=$?"; tail -2 $O/pytest_parity.txt | cut -c1-200
X="--no-cpu-baseline --extras= --latency-steps 0 --profile-steps 256"
for rep in 1 2 3; do for v in A B; do
  if [ $v = B ]; then export GUBER_HIP_LIB=$R/gubernator_amd/libguber_hip_v_pf2.so; else unset GUBER_HIP_LIB; fi
  timeout 120 python bench.py $X --dist uniform > $O/uniform_${v}_$rep.json 2> $O/uniform_${v}_$rep.err
  python -c "import json; d=json.load(open('$O/uniform_${v}_$rep.json')); print('uniform $v', round(d['value']/1e9,3), d['ms_per_step'], d.get('parity'), {k: v for k, v in d['roofline'].get('kernel_avg_us', {}).items() if 'multi' in k})"
done; done
for rep in 1 2; do for v in A B; do
  if [ $v = B ]; then export GUBER_HIP_LIB=$R/gubernator_amd/libguber_hip_v_pf2.so; else unset GUBER_HIP_LIB; fi
  timeout 120 python bench.py $X > $O/zipf_${v}_$rep.json 2> $O/zipf_${v}_$rep.err
  python -c "import json; d=json.load(open('$O/zipf_${v}_$rep.json')); print('zipf $v', round(d['value']/1e9,3), d['ms_per_step'], d.get('parity'))"
done; done
unset GUBER_HIP_LIB
