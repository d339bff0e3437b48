#!/bin/bash
# round 6, al: the kernel trace of gpu_r06_ak.sh shows BOTH decode streams on one hardware queue (73 % busy).  Does another priority for the second one, or more
# hardware queues (GPU_MAX_HW_QUEUES), separate them — and what does the rate say?  (laboratory build)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_am; mkdir -p $O; : > $O/al.txt
export TMPDIR=/tmp
K=10000000
mkdir -p /tmp/lablib; cp gubernator_amd/libguber_hip_lab.so /tmp/lablib/libguber_hip.so
queues() { python3 - "$1" <<'PY'
import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+'/*kernel_trace.csv')[0]
q=collections.defaultdict(lambda:[0,None,None,collections.Counter()])
for r in csv.DictReader(open(f)):
    k=r.get('Queue_Id'); a,b=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    e=q[k]; e[0]+=b-a; e[1]=a if e[1] is None else min(e[1],a); e[2]=b if e[2] is None else max(e[2],b); e[3][r['Kernel_Name'].split('(')[0].replace('guber::','')[:18]]+=1
print('   queues:', '; '.join('q%s %.0f%% %s'%(k,100*e[0]/max(1,e[2]-e[1]),list(dict(e[3].most_common(2)).keys())) for k,e in sorted(q.items()) if e[2]-e[1]>3e8))
PY
}
for v in "" "GUBER_WIRE_ROUTE_ON_ENGINES=1"; do
  rm -rf $O/trace
  env $v LD_LIBRARY_PATH=/tmp/lablib rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- tools/bench_pool_c 192 8 1000 $K 1.0 200 wire > $O/run.txt 2>&1
  echo "traced [$v]: $(grep '^pool:' $O/run.txt | sed -n 's/.*keys: *\([0-9.]*\) M decisions.*/\1 M\/s/p')" | tee -a $O/al.txt
  queues $O/trace | tee -a $O/al.txt
  for T in 64 128 192 256; do
    r=$(env $v LD_LIBRARY_PATH=/tmp/lablib timeout 120 tools/bench_pool_c $T 8 1000 $K 2.0 200 wire 2>&1 | grep "^pool:" | sed -n 's/.*keys: *\([0-9.]*\) M decisions.*p50 \([0-9.]*\) us p99 \([0-9.]*\) us, conservation: [0-9]* keys [0-9]* decisions \([0-9]*\) violations.*/\1 M\/s p50 \2 p99 \3 violations \4/p')
    echo "   untraced [$v] $T callers: $r" | tee -a $O/al.txt
  done
done
rm -rf $O/trace
