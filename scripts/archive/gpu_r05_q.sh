#!/bin/bash
# one table (the two-launch pipeline with claims): k_front verifies a key of <= 16 bytes against its two words in registers (no second
# trip to the key bytes), k_eval2 gets k_eval3's straight path for waves of nothing but common requests — against the build before,
# alternating on one box; the GPU parity suite on the new build first
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=$R/gpurun_out/r05_q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_parity.txt 2>&1; echo "pytest parity rc=$?"; tail -2 $O/pytest_parity.txt | cut -c1-200
X="--no-cpu-baseline --extras= --latency-steps 256 --profile-steps 256 --shards 1 --min-batches 1024 --steps 1024"
for rep in 1 2 3; do for v in A B; do
  if [ $v = A ]; then export GUBER_HIP_LIB=$R/gubernator_amd/libguber_hip_v_prev.so; else unset GUBER_HIP_LIB; fi
  timeout 120 python bench.py $X > $O/s1_${v}_$rep.json 2> $O/s1_${v}_$rep.err
  python -c "import json; d=json.load(open('$O/s1_${v}_$rep.json')); print('one table $v', round(d['value']/1e9,3), d['ms_per_step'], d['roofline'].get('kernel_avg_us'), 'idle p50', d['batch_latency']['idle']['p50'])"
done; done
for rep in 1 2; do for v in A B; do
  if [ $v = A ]; then export GUBER_HIP_LIB=$R/gubernator_amd/libguber_hip_v_prev.so; else unset GUBER_HIP_LIB; fi
  timeout 120 python bench.py --no-cpu-baseline --extras= --latency-steps 0 --profile-steps 256 > $O/h_${v}_$rep.json 2> $O/h_${v}_$rep.err
  python -c "import json; d=json.load(open('$O/h_${v}_$rep.json')); print('headline $v', round(d['value']/1e9,3), d['ms_per_step'])"
done; done
unset GUBER_HIP_LIB
