#!/bin/bash
# round 6: the front's own streams (1: routing + answers on one; 2: answers on their own; 3: two routing streams + answers) x generation size, full size
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
export GUBER_HIP_LIB=$PWD/gubernator_amd/libguber_hip_lab.so
for st in 1 2 3; do for gb in 8 12 16; do for d in 4 8; do
  GUBER_BENCH_FRONT_DEPTH=$d GUBER_FRONT_STREAMS=$st timeout 600 python tools/front_probe.py 10000000 $gb $((1024 / gb)) 2>&1 | grep "in one call" | sed "s/^/streams $st depth $d: /" | cut -c1-200 | tee -a gpurun_out/r06_d_front_streams.txt
done; done; done
