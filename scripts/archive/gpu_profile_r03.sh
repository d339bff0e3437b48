#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of the DEFAULT bench command (CPU legs and extras off: same GPU work as the
# headline), a one-table trace, and PMC passes (each counter group in its own run, no tracing) on the non-replayed stream.
# usage: gpu_profile_r03.sh <tag> [pmc_batches]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03}
NB=${2:-512}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
HEAD="--no-cpu-baseline --extras="
rm -rf $O/trace_fused
echo "python bench.py $HEAD" > $O/trace_fused.cmd
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_fused -o t -- python $R/bench.py $HEAD > $O/trace_fused.log 2>&1; echo "trace fused rc=$?"
rm -rf $O/trace_s1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_s1 -o t -- python $R/bench.py $HEAD --shards 1 --min-batches 1024 --steps 1024 > $O/trace_s1.log 2>&1; echo "trace S=1 rc=$?"
PARGS="$HEAD --min-batches $NB --steps $NB --warmup 8 --profile-steps 0 --latency-steps 0"
PMC_SHARDS=${PMC_SHARDS:-"1 12"}
for ctr in ${PMC_SETS:-FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum+TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum+TCC_ATOMIC_sum TCC_HIT_sum+TCC_MISS_sum}; do
  ctr=$(echo $ctr | tr "+" " ")
  n=$(echo $ctr | tr ' ' '+')
  for s in $PMC_SHARDS; do
    rm -rf $O/pmc_s${s}_$n
    timeout 900 rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_s${s}_$n -o pmc -- python $R/bench.py $PARGS --shards $s > $O/pmc_s${s}_$n.log 2>&1; echo "pmc S=$s $ctr rc=$?"
  done
done
cd $R && python tools/summarize_r03.py $TAG $NB
