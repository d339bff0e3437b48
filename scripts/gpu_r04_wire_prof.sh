cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_wire -o w -- python -m pytest $R/tests/test_gpu_wire_dev.py -q -k throughput -s > $R/gpurun_out/r04_wire.log 2>&1
f=$(find $R/gpurun_out/r04_wire -name "*kernel_stats.csv" | head -1); grep -i "wire" $f | cut -c1-200; cp $f $R/gpurun_out/r04_wire_kernel_stats.csv; rm -rf $R/gpurun_out/r04_wire
