#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=$R/gpurun_out/r05_d; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_bench_multi.py > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.txt | cut -c1-300
bash scripts/gpu_r05_ab.sh r05_d_ab libguber_hip_v_prediet.so default 3
bash scripts/gpu_r05_sq.sh r05_d_sq_ep0 GUBER_FUSE_EP=0 2>&1 | grep "per wave\|^k_" 
timeout 300 python -m pytest tests/test_gpu_bench_multi.py -m gpu -q -x -s > $O/pytest_multi.txt 2>&1; echo "pytest multi rc=$?"; tail -30 $O/pytest_multi.txt | cut -c1-400
