#!/bin/bash
# round 6, y: the answers' last hop of generation g launched after the evaluations of generation g + 1 (laboratory knob GUBER_FRONT_OUT_DELAY): does the
# engine stream that carries it stop waiting for the other streams' evaluations of g?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/r06_y_out_delay_ab.txt; : > $O
ARGS="--no-cpu-baseline --extras= --min-batches 1024 --steps 1024 --profile-steps 0 --latency-steps 0 --headline routed"
export GUBER_HIP_LIB=$PWD/gubernator_amd/libguber_hip_lab.so
for rep in 1 2 3; do for v in 0 1 2; do for depth in 4 6; do
  val=$(GUBER_FRONT_OUT_DELAY=$v GUBER_BENCH_FRONT_DEPTH=$depth timeout 600 python bench.py $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e9,3), str(d.get('parity'))[:30])")
  echo "rep $rep out_delay $v depth $depth: $val" | tee -a $O
done; done; done
