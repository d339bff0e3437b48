#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r02_final8; mkdir -p $O
timeout 100 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.txt | cut -c1-200
