#!/bin/bash
# Runs on the GPU box at the end of a round: full GPU test suite, the three bench lines, rocprofv3 kernel stats + PMC
# traffic (scripts/gpu_profile.sh), hardware-counter attribution passes (scripts/gpu_pmc.sh), phase timestamps.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; tail -3 gpurun_out/pytest_gpu.txt
timeout 600 python bench.py > gpurun_out/bench_token.json 2> gpurun_out/bench_token.err; echo "token rc=$?"; cat gpurun_out/bench_token.json
timeout 300 python bench.py --algo leaky --no-cpu-baseline > gpurun_out/bench_leaky.json 2>/dev/null; echo "leaky rc=$?"; cat gpurun_out/bench_leaky.json
timeout 300 python bench.py --dist uniform --no-cpu-baseline > gpurun_out/bench_uniform.json 2>/dev/null; echo "uniform rc=$?"; cat gpurun_out/bench_uniform.json
timeout 300 python bench.py --shards 1 --no-cpu-baseline > gpurun_out/bench_token_s1.json 2>/dev/null; echo "s1 rc=$?"; cat gpurun_out/bench_token_s1.json
timeout 300 python bench.py --global-sync 16 --steps 64 --no-cpu-baseline --profile-steps 0 > gpurun_out/bench_global.json 2>/dev/null; echo "global rc=$?"; cat gpurun_out/bench_global.json
./scripts/gpu_profile.sh r01_final 2>&1 | tail -12
./scripts/gpu_pmc.sh "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum" 2>&1 | tail -6
./tools/phase_timing.sh 2>&1 | tail -3
