#!/bin/bash
# fused multi-engine launches: parity test, then rate with / without fusing
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r02_o; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "routed or back_to_back or adversarial" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt | cut -c1-400
digest='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]/1e9,3), "G/s", d["ms_per_step"], d["roofline"]["kernel_avg_us"], d.get("parity")); print("   ", d["timed_region"]["shard_streams"])'
for fuse in 1 0; do
  if [ $fuse = 0 ]; then export GUBER_NO_FUSE=1; else unset GUBER_NO_FUSE; fi
  for S in 2 4 8 16; do
    echo "== zipf S=$S dispatch=one fuse=$fuse" | tee -a $O/fuse.txt
    timeout 300 python bench.py --no-cpu-baseline --extras "" --shards $S --dispatch one 2>$O/err.txt | python -c "$digest" | tee -a $O/fuse.txt
    tail -2 $O/err.txt | cut -c1-300
  done
done
unset GUBER_NO_FUSE
for d in uniform; do
  for S in 4 8; do
    echo "== $d S=$S dispatch=one fuse=1" | tee -a $O/fuse.txt
    timeout 300 python bench.py --no-cpu-baseline --extras "" --shards $S --dist $d --dispatch one 2>$O/err.txt | python -c "$digest" | tee -a $O/fuse.txt
  done
done
echo "== leaky S=4 dispatch=one" | tee -a $O/fuse.txt
timeout 300 python bench.py --no-cpu-baseline --extras "" --shards 4 --algo leaky --dispatch one 2>$O/err.txt | python -c "$digest" | tee -a $O/fuse.txt
