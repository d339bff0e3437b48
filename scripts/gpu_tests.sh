#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q ${1:-} > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -40 gpurun_out/pytest_gpu.txt | cut -c1-400
