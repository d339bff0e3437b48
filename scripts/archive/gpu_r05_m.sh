#!/bin/bash
# where k_wire_fill's 16 us go: measurement builds that leave out the parse / the HashKey rows / the columns (kernel trace, by shape)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=$R/gpurun_out/r05_m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in default wf_NO_PARSE wf_NO_ROWS wf_NO_COLS; do
  lib=$R/gubernator_amd/libguber_hip_v_$v.so; [ $v = default ] && lib=$R/gubernator_amd/libguber_hip.so
  GUBER_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$v -o t -- python -m pytest $R/tests/test_gpu_wire_dev.py -m gpu -q -k "report_throughput" > $O/trace_$v.log 2>&1; echo "$v trace rc=$?"
  t=$(find $O/tr_$v -name "*kernel_trace.csv" | head -1)
  python $R/tools/wire_trace_by_shape.py "$t" | tee $O/by_shape_$v.txt
  rm -rf $O/tr_$v
done
