#!/bin/bash
# A/B of k_front's warm-up workgroups (GUBER_NO_WARM=1 turns them off), same box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r02_l; mkdir -p $O
digest='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]/1e9,3), "G/s", d["ms_per_step"], d["roofline"]["kernel_avg_us"], {k:(round(v["value"]/1e9,3) if isinstance(v,dict) and "value" in v else None) for k,v in d.items() if k in ("leaky","shards_1","uniform")}, (d.get("shards_1") or {}).get("batch_latency"))'
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt | cut -c1-300
for rep in 1 2; do
for w in warm nowarm; do
  if [ $w = nowarm ]; then export GUBER_NO_WARM=1; else unset GUBER_NO_WARM; fi
  for S in 1 4 8; do
    echo "== $w S=$S rep=$rep"
    timeout 300 python bench.py --no-cpu-baseline --extras "" --shards $S 2>$O/err.txt | python -c "$digest" | tee -a $O/ab.txt
  done
done
done
unset GUBER_NO_WARM
for w in warm nowarm; do
  if [ $w = nowarm ]; then export GUBER_NO_WARM=1; else unset GUBER_NO_WARM; fi
  echo "== timing build $w: bench.py --shards 1" | tee -a $O/phase_timing.txt
  GUBER_HIP_LIB=$R/gubernator_amd/libguber_hip_timing.so timeout 300 python bench.py --no-cpu-baseline --extras "" --shards 1 --profile-steps 0 2>&1 >/dev/null | grep -A16 "phase timing" | tee -a $O/phase_timing.txt
done
