#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r02_s; mkdir -p $O
for depth in 2 3; do
  echo "== end_to_end depth=$depth" | tee -a $O/e2e.txt
  GUBER_BENCH_E2E_DEPTH=$depth timeout 400 python bench.py --no-cpu-baseline --extras end_to_end --profile-steps 0 --steps 32 --min-ms 30 2>$O/e2e.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['end_to_end']; d.pop('workload'); print(json.dumps(d))" | tee -a $O/e2e.txt; tail -2 $O/e2e.err | cut -c1-300
done
cd /tmp && export TMPDIR=/tmp
GUBER_BENCH_E2E_DEPTH=3 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $R/$O/prof -o t -- python $R/bench.py --no-cpu-baseline --extras end_to_end --profile-steps 0 --steps 8 --min-ms 10 --warmup 2 > $R/$O/prof.log 2>&1
ls $R/$O/prof | head; find $R/$O/prof -name "*memory_copy_trace.csv" | head -1 | xargs -I{} sh -c 'tail -40 {}' | cut -c1-250
