#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel traces of the DEFAULT bench command (CPU legs and extras off: same GPU work as the headline),
# of one table with the default policy (claims pipeline) and with GUBER_PIPELINE=part, and PMC passes (each counter group in its
# own run, no tracing) on the non-replayed stream, and the SQ counter passes behind roofline.issue (scripts/gpu_r05_sq.sh).
# The FETCH / WRITE passes over the fused configuration run with GUBER_FUSE_EP=0: the same kernels' work, launched one by one, so
# that the bytes are per kernel (the default packs k_eval3 + the next k_part into one launch).      usage: gpu_profile_r05.sh <tag> [pmc_batches]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05}
NB=${2:-256}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
HEAD="--no-cpu-baseline --extras="
rm -rf $O/trace_fused
echo "python bench.py $HEAD" > $O/trace_fused.cmd
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_fused -o t -- python $R/bench.py $HEAD > $O/trace_fused.log 2>&1; echo "trace fused rc=$?"
rm -rf $O/trace_s1 $O/trace_s1_part
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_s1 -o t -- python $R/bench.py $HEAD --shards 1 --min-batches 1024 --steps 1024 > $O/trace_s1.log 2>&1; echo "trace S=1 rc=$?"
GUBER_PIPELINE=part timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_s1_part -o t -- python $R/bench.py $HEAD --shards 1 --min-batches 1024 --steps 1024 > $O/trace_s1_part.log 2>&1; echo "trace S=1 part rc=$?"
PARGS="$HEAD --min-batches $NB --steps $NB --warmup 8 --profile-steps 0 --latency-steps 0"
for ctr in ${PMC_SETS:-FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum+TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum+TCC_ATOMIC_sum TCC_HIT_sum+TCC_MISS_sum}; do
  c=$(echo $ctr | tr "+" " ")
  rm -rf $O/pmc_s12_$ctr
  GUBER_FUSE_EP=0 timeout 900 rocprofv3 --pmc $c --output-format csv -d $O/pmc_s12_$ctr -o pmc -- python $R/bench.py $PARGS --shards 12 > $O/pmc_s12_$ctr.log 2>&1; echo "pmc S=12 $c rc=$?"
done
for ctr in ${PMC_SETS_S1:-FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum+TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum+TCC_ATOMIC_sum}; do
  c=$(echo $ctr | tr "+" " ")
  rm -rf $O/pmc_s1_$ctr $O/pmc_s1p_$ctr
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $O/pmc_s1_$ctr -o pmc -- python $R/bench.py $PARGS --shards 1 > $O/pmc_s1_$ctr.log 2>&1; echo "pmc S=1 claims $c rc=$?"
  GUBER_PIPELINE=part timeout 900 rocprofv3 --pmc $c --output-format csv -d $O/pmc_s1p_$ctr -o pmc -- python $R/bench.py $PARGS --shards 1 > $O/pmc_s1p_$ctr.log 2>&1; echo "pmc S=1 part $c rc=$?"
done
cd $R && bash scripts/gpu_r05_sq.sh ${TAG}_sq GUBER_FUSE_EP=0 > $O/sq.log 2>&1; cp $R/gpurun_out/${TAG}_sq/sq_counters.txt $O/${TAG}_sq_counters.txt; cp $R/gpurun_out/${TAG}_sq/sq_counters.json $O/sq_counters.json
cd $R && python tools/summarize_r05.py $TAG $NB; echo "summarize rc=$?"
# keep what is committed small: drop the raw rocprofv3 trees, keep the summaries, the logs' bench lines and the kernel stats
for d in trace_fused trace_s1 trace_s1_part; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${TAG}_${d}_kernel_stats.csv; done
rm -rf $O/trace_fused $O/trace_s1 $O/trace_s1_part $O/pmc_s1*_*/ $O/pmc_s12_*/ 2>/dev/null
ls $O
