#!/bin/bash
# a soak of the device wire decoder's new chain walk against the host transcoder: the fuzz test and the many-windows test under 24 seeds
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=$R/gpurun_out/r05_p; mkdir -p $O
bad=0
for seed in $(seq 100 123); do
  GUBER_WIRE_FUZZ_SEED=$seed timeout 300 python -m pytest tests/test_gpu_wire_dev.py -m gpu -q -x -k "fuzzed or many_windows" > $O/seed_$seed.txt 2>&1 || { bad=$((bad+1)); echo "seed $seed FAILED"; tail -20 $O/seed_$seed.txt; }
done
echo "wire soak: 24 seeds, $bad failed"
