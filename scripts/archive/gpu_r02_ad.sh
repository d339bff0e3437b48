#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r02_ad; mkdir -p $O
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --router slots --cpu-threads 32 --cpu-seconds 1 --extras "" > $O/out.json 2> $O/err.txt; echo rc=$?
grep -i "parity" $O/err.txt | cut -c1-1500 | head -12
