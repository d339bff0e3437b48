#!/bin/bash
# round 6, as: the wire files of the GPU suite, then the kernel trace of the payload stage in its FINAL form (k_wire_enc in the last hop's place) under 192 callers (8 tables): per-kernel durations and how busy each hardware queue is
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_as; mkdir -p $O
export TMPDIR=/tmp
K=10000000
timeout 900 python -m pytest tests/test_gpu_wire_pool.py tests/test_gpu_wire_dev.py tests/test_gpu_host_layer.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2 > $O/tests.txt; cat $O/tests.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o wire192 -- tools/bench_pool_c 192 8 1000 $K 1.0 200 wire > $O/run.txt 2>&1
grep -v amdgpu.ids $O/run.txt | grep "pool" > $O/summary.txt
cp $O/trace/*kernel_stats.csv $O/wire192_kernel_stats.csv
python3 - >> $O/summary.txt <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/r06_as/trace/*kernel_stats.csv')[0]
print("kernel                                              calls   avg us   min us   max us   share")
for r in list(csv.DictReader(open(f)))[:16]:
    print(r['Name'][:50].ljust(50), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(8), ('%.1f'%(float(r['MinNs'])/1e3)).rjust(8), ('%.1f'%(float(r['MaxNs'])/1e3)).rjust(8), r['Percentage'].rjust(7))
f=glob.glob('gpurun_out/r06_as/trace/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
q=collections.defaultdict(lambda:[0,None,None,collections.Counter()])
for r in rows:
    k=r.get('Queue_Id') or r.get('Stream_Id')
    a,b=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    e=q[k]; e[0]+=b-a; e[1]=a if e[1] is None else min(e[1],a); e[2]=b if e[2] is None else max(e[2],b); e[3][r['Kernel_Name'].split('(')[0].replace('guber::','')[:22]]+=1
for k,e in sorted(q.items()):
    if e[2]-e[1] > 5e8: print('hardware queue',k,'busy %.0f %% of %.2f s:'%(100*e[0]/max(1,e[2]-e[1]),(e[2]-e[1])/1e9), dict(e[3].most_common(5)))
PY
rm -rf $O/trace
cat $O/summary.txt
