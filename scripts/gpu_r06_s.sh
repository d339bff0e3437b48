#!/bin/bash
# round 6, s: where the payload stage's time goes at 256 callers: the flusher's own time per call, and the kernel trace
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_s; mkdir -p $O
export TMPDIR=/tmp
K=10000000
timeout 120 tools/bench_pool_c 256 8 1000 $K 2.0 200 wire 2>&1 | grep -v amdgpu.ids >> $O/pool_wire.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o wire256 -- tools/bench_pool_c 256 8 1000 $K 1.0 200 wire > $O/run.txt 2>&1
grep -v amdgpu.ids $O/run.txt | grep "pool" >> $O/pool_wire.txt
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r06_s/trace/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:18]:
    print(r['Name'][:50].ljust(50), r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'], r['Percentage'])
PY
cp $O/trace/*kernel_stats.csv $O/wire256_kernel_stats.csv
python3 - <<'PY'
# queue occupancy: per stream (queue id) busy time / span
import csv,glob,collections
f=glob.glob('gpurun_out/r06_s/trace/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
q=collections.defaultdict(lambda:[0,None,None,collections.Counter()])
for r in rows:
    k=r.get('Queue_Id') or r.get('Stream_Id')
    a,b=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    e=q[k]; e[0]+=b-a; e[1]=a if e[1] is None else min(e[1],a); e[2]=b if e[2] is None else max(e[2],b); e[3][r['Kernel_Name'][:24]]+=1
for k,e in q.items():
    print('queue',k,'busy %.1f%%'%(100*e[0]/max(1,e[2]-e[1])),'span %.2fs'%((e[2]-e[1])/1e9), dict(e[3].most_common(4)))
PY
rm -rf $O/trace
cat $O/pool_wire.txt
