#!/bin/bash
# round 3, call A: GPU tests, the first non-replayed bench line, kernel traces + FETCH/WRITE counters (one table)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_a
mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench.err
PMC_SHARDS="1" PMC_SETS="FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum+TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum+TCC_ATOMIC_sum" bash scripts/gpu_profile_r03.sh r03_a 256 > $O/profile.log 2>&1
tail -40 $O/profile.log
