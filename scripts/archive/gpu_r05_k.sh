#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=$R/gpurun_out/r05_k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_parity.txt 2>&1; echo "pytest parity rc=$?"; tail -2 $O/pytest_parity.txt | cut -c1-200
bash scripts/gpu_r05_ab.sh r05_k_ab libguber_hip_v_prev.so default 2
X="--no-cpu-baseline --extras= --latency-steps 0 --profile-steps 256 --algo leaky"
for rep in 1 2; do for v in A B; do
  if [ $v = A ]; then export GUBER_HIP_LIB=$R/gubernator_amd/libguber_hip_v_prev.so; else unset GUBER_HIP_LIB; fi
  timeout 120 python bench.py $X > $O/leaky_${v}_$rep.json 2> $O/leaky_${v}_$rep.err
  python -c "import json; d=json.load(open('$O/leaky_${v}_$rep.json')); print('leaky $v', round(d['value']/1e9,3), d['ms_per_step'])"
done; done
unset GUBER_HIP_LIB
