#!/bin/bash
# round 6: does the system-scope release of the counter snapshots (k_part / k_front tile 0: __threadfence_system + release store = the XCD's L2 written back)
# cost the launch it rides on?  A build whose stamp is a relaxed store (measurement only) against the product, alternating on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/r06_k_snapshot_release_ab.txt; : > $O
ARGS="--no-cpu-baseline --extras= --min-batches 1024 --steps 1024 --profile-steps 0 --latency-steps 0"
for rep in 1 2 3; do for v in default snapweak; do for hl in routed presplit; do
  if [ $v = default ]; then unset GUBER_HIP_LIB; else export GUBER_HIP_LIB=$PWD/gubernator_amd/libguber_hip_v_snapweak.so; fi
  val=$(timeout 600 python bench.py $ARGS --headline $hl 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value']/1e9,3))")
  echo "rep $rep $v $hl $val" | tee -a $O
done; done; done
