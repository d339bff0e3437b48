#!/bin/bash
# the GLOBAL leg's convergence check: two processes through the RCCL call sequence against two logical ranks in one process, same
# keys / steps; the probe's answers as sums (the streams are seeded: every run should say the same)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=$R/gpurun_out/r05_v; mkdir -p $O
export GUBER_BENCH_GLOBAL_SUMS=1
A="--global-sync 8 --keys 200000 --steps 32 --warmup 8"
for i in 1 2 3 4 5 6; do
  GUBER_RCCL_LIB=$R/tests/hostsim/libfake_rccl.so timeout 200 python bench.py --gpus 2 --one-device --backend gloo $A > $O/two_$i.json 2> $O/two_$i.err; echo "two processes run $i rc=$?"; grep "global leg\]" $O/two_$i.err | cut -c1-300
done
for i in 1 2 3 4; do
  timeout 200 python bench.py --gpus 1 $A > $O/one_$i.json 2> $O/one_$i.err; echo "one process, two logical ranks run $i rc=$?"; grep "global leg\]" $O/one_$i.err | cut -c1-300
done
