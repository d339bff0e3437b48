#!/bin/bash
# round 4, step n: what random bucket accesses sustain (tools/random_access), the retry-reorder test, the pool's GPU tests with the device routing
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; O=gpurun_out/r04_n
timeout 300 tools/random_access 2>&1 | grep -v amdgpu.ids | tee ${O}_random_access.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "retries_may_run or faulty_ranks or device_routes" 2>&1 | tail -5 | cut -c1-300
GUBER_POOL_DEVROUTE=1 timeout 900 python -m pytest tests/test_gpu_host_layer.py -m gpu -q -x > ${O}_pytest_host_layer_devroute.txt 2>&1; echo "host layer with device routing rc=$?"; grep -n "passed\|failed" ${O}_pytest_host_layer_devroute.txt | cut -c1-300
