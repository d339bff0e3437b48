#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r02_aa; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt | cut -c1-400
timeout 400 python bench.py --no-cpu-baseline --extras end_to_end,pool --profile-steps 0 --steps 32 --min-ms 30 2>$O/e2e.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['end_to_end']; e.pop('workload'); print(json.dumps(e)); p=d['pool']; p.pop('workload'); print(json.dumps(p))" | tee $O/e2e.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
