#!/bin/bash
# round 6: k_eval2 / k_eval3 / k_evalpart held to 3 waves per SIMD (152 VGPRs, no scratch) against the default 4 (128 VGPRs, 30 - 40 spilled registers, 72 - 88 B of scratch per lane):
# alternating builds on one box, both arrangements
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/r06_h_eval_waves_ab.txt; : > $O
ARGS="--no-cpu-baseline --extras= --min-batches 1024 --steps 1024 --profile-steps 0 --latency-steps 0"
for rep in 1 2 3; do for v in default ew3; do for hl in routed presplit; do
  if [ $v = default ]; then unset GUBER_HIP_LIB; else export GUBER_HIP_LIB=$PWD/gubernator_amd/libguber_hip_v_ew3.so; fi
  val=$(timeout 600 python bench.py $ARGS --headline $hl 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value']/1e9,3))")
  echo "rep $rep $v $hl $val" | tee -a $O
done; done; done
unset GUBER_HIP_LIB
timeout 600 python -m pytest tests/test_gpu_host_layer.py -q -x 2>&1 | tail -2
