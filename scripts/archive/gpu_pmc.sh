#!/bin/bash
# Runs on the GPU box: hardware-counter passes (rocprofv3 --pmc, one small group per pass, no tracing) over
# bench.py with one logical shard, to attribute the time of k_front / k_eval2.  Output: gpurun_out/pmc2_<group>/.
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3-avail list > $R/gpurun_out/pmc_avail.txt 2>&1 || rocprofv3 --list-avail > $R/gpurun_out/pmc_avail.txt 2>&1
ARGS="--no-cpu-baseline --shards 1 --steps 32 --warmup 4 --profile-steps 0"
i=0
for grp in "$@"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc2_$i
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc2_$i -o pmc -- python $R/bench.py $ARGS > $R/gpurun_out/pmc2_$i.log 2>&1
  echo "pass $i [$grp] rc=$?"
done
