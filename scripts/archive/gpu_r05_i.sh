#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=$R/gpurun_out/r05_i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_parity.txt 2>&1; echo "pytest parity rc=$?"; tail -2 $O/pytest_parity.txt | cut -c1-200
bash scripts/gpu_r05_ab.sh r05_i_ab libguber_hip_v_prev.so default 3
