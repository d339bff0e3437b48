#!/bin/bash
# round 6, ae: the routed headline against the generation's size — the shares' PIECES were 43 000 requests in every arrangement measured so far (8, 16, 24 batches
# per generation over 12 tables); pieces near the pipelines' 65 536?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
O=gpurun_out/r06_ae_gen_size.txt; : > $O
ARGS="--no-cpu-baseline --extras= --min-batches 1024 --steps 1024 --profile-steps 0 --latency-steps 0 --headline routed"
for rep in 1 2; do for gb in 10 11 12 16 20 22 32; do
  val=$(timeout 600 python bench.py $ARGS --gen-batches $gb 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e9,3))")
  echo "rep $rep gen-batches $gb: $val G/s" | tee -a $O
done; done
