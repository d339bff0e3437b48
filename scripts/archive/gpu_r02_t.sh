#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r02_t; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/launch_rate tools/launch_rate.hip -lpthread 2>/dev/null && timeout 300 /tmp/launch_rate | tail -6 | tee $O/host_call_cost.txt
for inplace in 0 1; do
  if [ $inplace = 1 ]; then export GUBER_STAGE_OUT_INPLACE=1; else unset GUBER_STAGE_OUT_INPLACE; fi
  echo "== end_to_end depth=2 out_inplace=$inplace" | tee -a $O/e2e.txt
  GUBER_BENCH_E2E_DEPTH=2 timeout 400 python bench.py --no-cpu-baseline --extras end_to_end --profile-steps 0 --steps 32 --min-ms 30 2>$O/e2e.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['end_to_end']; d.pop('workload'); print(json.dumps(d))" | tee -a $O/e2e.txt; tail -2 $O/e2e.err | cut -c1-300
done
