#!/bin/bash
# round 6, aw: the whole GPU suite, smoke() and the driver's bench command on the final tree
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_aw; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/tests.txt
cat $O/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > $O/smoke.txt
cat $O/smoke.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err
grep real $O/bench.err
tail -c 600 $O/bench.err | grep -v amdgpu.ids | tail -3
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06_aw/bench.json') if l.startswith('{')][-1])
print('value', d['value'], 'frac', d['roofline']['frac'], 'routed/presplit', d.get('routed_over_presplit'))
for k,v in d.get('extras',{}).items():
    if isinstance(v,dict):
        print(' ',k, v.get('value'), v.get('parity','')[:40] if isinstance(v.get('parity'),str) else '')
p=d.get('extras',{}).get('pool',{})
for k,v in p.items():
    if isinstance(v,dict) and 'caller_threads' in v: print('   pool', k, v.get('value'), v.get('rpc_latency_us'), (v.get('parity') or '')[:30])
PY
