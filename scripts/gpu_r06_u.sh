#!/bin/bash
# round 6, u: what the front thread's 55 us per stage are made of (the laboratory build's dispatch profile)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_u; mkdir -p $O
K=10000000
LD_PRELOAD= GUBER_DISPATCH_PROFILE=1 LD_LIBRARY_PATH=/tmp/lablib:$LD_LIBRARY_PATH true
mkdir -p /tmp/lablib; cp gubernator_amd/libguber_hip_lab.so /tmp/lablib/libguber_hip.so
GUBER_DISPATCH_PROFILE=1 LD_LIBRARY_PATH=/tmp/lablib timeout 120 tools/bench_pool_c 64 8 1000 $K 1.0 200 wire > $O/run.txt 2>$O/err.txt
grep -v amdgpu.ids $O/run.txt
python3 - <<'PY'
import re,collections
a=collections.defaultdict(list)
for l in open('gpurun_out/r06_u/err.txt'):
    if l.startswith('[front]'):
        m=re.findall(r'([a-z\' ]+) ([0-9.]+)(?: us)?', l.split('per generation:')[1])
        for k,v in re.findall(r"([a-zA-Z' ]+?) ([0-9]+\.[0-9]+)", l.split('per generation:')[1]): a['front:'+k.strip()].append(float(v))
    elif l.startswith('[front dispatch]'):
        mm=re.match(r'\[front dispatch\] (\d+) batches in (\d+) groups', l)
        a['disp:batches'].append(float(mm.group(1))); a['disp:groups'].append(float(mm.group(2)))
        for k,v in re.findall(r"([a-zA-Z+\- ]+?) ([0-9]+\.[0-9]+)", l.split('per batch:')[1]): a['disp/batch:'+k.strip()].append(float(v))
for k,v in a.items(): print(k.ljust(40), 'n', len(v), 'mean %.2f'%(sum(v)/len(v)))
PY
rm -f $O/err.txt
