#!/bin/bash
# round 6, n: kernel trace of the payload stage under 128 callers
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_n; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o wire128 -- tools/bench_pool_c 128 8 1000 10000000 1.0 200 wire > $O/run.txt 2>&1
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/wire128_kernel_stats.csv
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r06_n/wire128_kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:30]:
    print(r['Name'][:60].ljust(60), r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['MinNs'], r['MaxNs'], r['Percentage'])
PY
grep -v amdgpu.ids $O/run.txt | tail -3
find $O/trace -name "*kernel_trace.csv" -size +60M -delete
