#!/bin/bash
# round 6, aq: kernel trace of the payload stage with the responses encoded on the device and on the host (64 and 192 callers): what k_wire_enc costs a stage's chain
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_aq; mkdir -p $O; : > $O/aq.txt
export TMPDIR=/tmp
K=10000000
mkdir -p /tmp/lablib; cp gubernator_amd/libguber_hip_lab.so /tmp/lablib/libguber_hip.so
stats() { python3 - "$1" <<'PY'
import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv', recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r['Kernel_Name'].split('(')[0].replace('guber::','')].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    v.sort(); print('   %-22s n=%6d avg=%7.2f us p50=%7.2f max=%8.2f total=%8.1f ms'%(k[:22],len(v),sum(v)/len(v),v[len(v)//2],v[-1],sum(v)/1e3))
PY
}
for T in 64 192; do for v in "GUBER_WIRE_HOST_ENCODE=0" "GUBER_WIRE_HOST_ENCODE=1"; do
  rm -rf $O/trace
  env $v LD_LIBRARY_PATH=/tmp/lablib rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- tools/bench_pool_c $T 8 1000 $K 1.0 200 wire > $O/run.txt 2>&1
  echo "traced [$v] $T callers: $(grep '^pool:' $O/run.txt | sed -n 's/.*keys: *\([0-9.]*\) M decisions.*/\1 M\/s/p'); $(grep '^wire pool' $O/run.txt | sed -n 's/.*per stage: \(.*\)/\1/p')" | tee -a $O/aq.txt
  stats $O/trace | tee -a $O/aq.txt
done; done
rm -rf $O/trace
