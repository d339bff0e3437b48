#!/bin/bash
# the GLOBAL leg with two processes: the engines' own statistics (eviction pre-passes, waits for a snapshot) beside the probe's sums
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=$R/gpurun_out/r05_w; mkdir -p $O
export GUBER_BENCH_GLOBAL_SUMS=1
A="--global-sync 8 --keys 200000 --steps 32 --warmup 8"
for i in $(seq 1 ${1:-8}); do
  GUBER_RCCL_LIB=$R/tests/hostsim/libfake_rccl.so timeout 200 python bench.py --gpus 2 --one-device --backend gloo $A > $O/two_$i.json 2> $O/two_$i.err; echo "two processes run $i rc=$?"; grep "global leg\] rank . sums\|global leg\] rank .: sums\|\[engine" $O/two_$i.err | cut -c1-420
done
