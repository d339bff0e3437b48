#!/bin/bash
# round 6: the key's hash travels from k_fr_count to k_part (no second XXH64 per request) — A/B on one box (laboratory build, the switch only decides whether k_part takes it)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/r06_j_pass_hash_ab.txt; : > $O
timeout 300 python -m pytest tests/test_gpu_front.py -x -q 2>&1 | tail -2
export GUBER_HIP_LIB=$PWD/gubernator_amd/libguber_hip_lab.so
ARGS="--no-cpu-baseline --extras= --min-batches 1024 --steps 1024 --profile-steps 0 --latency-steps 0"
for rep in 1 2 3; do for v in 0 1; do
  val=$(GUBER_FRONT_PASS_HASH=$v timeout 600 python bench.py $ARGS 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value']/1e9,3))")
  echo "rep $rep pass_hash $v routed $val" | tee -a $O
done; done
