#!/bin/bash
# Runs on the GPU box: the rocprofv3 evidence behind round 6's bench line.
#   1. kernel traces (--kernel-trace --stats) of the DEFAULT bench command (the routed headline; CPU legs and extras off: same GPU work), of the
#      pre-split arrangement, and of one table;
#   2. PMC passes (each counter group in its own run, no tracing) over the routed arrangement: FETCH_SIZE / WRITE_SIZE (HBM-side bytes) and the
#      SQ counters behind roofline.issue.  They run on the laboratory build with GUBER_FUSE_EP=0 so that k_eval3 and k_part are launches of
#      their own (the default packs a generation's k_eval3 with the next one's k_part): the same kernels' work, counted one by one.
#   usage: gpu_profile_r06.sh <tag> [pmc_batches]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}
NB=${2:-256}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
HEAD="--no-cpu-baseline --extras="
for cfg in "routed:" "presplit:--headline presplit" "shards_1:--headline presplit --shards 1 --min-batches 1024 --steps 1024"; do
  name=${cfg%%:*}; extra=${cfg#*:}
  rm -rf $O/trace_$name
  echo "python bench.py $HEAD $extra" > $O/trace_$name.cmd
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -o t -- python $R/bench.py $HEAD $extra > $O/trace_$name.log 2>&1; echo "trace $name rc=$?"
done
PARGS="$HEAD --min-batches $NB --steps $NB --warmup 16 --profile-steps 0 --latency-steps 0"
echo "GUBER_HIP_LIB=gubernator_amd/libguber_hip_lab.so GUBER_FUSE_EP=0 python bench.py $PARGS" > $O/pmc.cmd
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf $O/pmc_$i
  GUBER_HIP_LIB=$R/gubernator_amd/libguber_hip_lab.so GUBER_FUSE_EP=0 timeout 600 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$i -o pmc -- python $R/bench.py $PARGS > $O/pmc_$i.log 2>&1
  echo "pmc pass $i [$grp] rc=$?"; grep -i "error\|invalid\|not found" $O/pmc_$i.log | head -3 | cut -c1-200
done
cd $R && python tools/summarize_r06.py $TAG $NB; echo "summarize rc=$?"
# keep what is committed small: drop the raw rocprofv3 trees, keep the summaries, the logs' bench lines and the kernel stats
for d in trace_routed trace_presplit trace_shards_1; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${TAG}_${d}_kernel_stats.csv; done
rm -rf $O/trace_routed $O/trace_presplit $O/trace_shards_1 $O/pmc_[0-9]*/ 2>/dev/null
ls $O
