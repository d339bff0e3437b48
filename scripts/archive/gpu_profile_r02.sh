#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace (durations isolated at S=1 and overlapped at S=4) and PMC passes
# (FETCH_SIZE, WRITE_SIZE, L2 hit/miss + fabric requests; each in its own run) for bench.py.  usage: gpu_profile_r02.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --extras= --steps 64 --warmup 8 --min-ms 40 --profile-steps 0"
for s in 1 4; do
  rm -rf $O/trace_s$s
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_s$s -o t -- python $R/bench.py $ARGS --shards $s --dispatch threads > $O/trace_s$s.log 2>&1; echo "trace S=$s rc=$?"
done
rm -rf $O/trace_fused
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_fused -o t -- python $R/bench.py $ARGS --shards 12 --streams 3 --dispatch one > $O/trace_fused.log 2>&1; echo "trace fused rc=$?"
for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum"; do
  n=$(echo $ctr | tr ' ' '+')
  rm -rf $O/pmc_$n
  timeout 600 rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_$n -o pmc -- python $R/bench.py $ARGS --shards 1 > $O/pmc_$n.log 2>&1; echo "pmc $ctr rc=$?"
done
cd $R && python tools/summarize_r02.py $TAG
