#!/bin/bash
# A/B of builds on ONE box, alternating: gpu_r05_ab.sh <tag> <libA> <libB> [reps]     (libs: paths under gubernator_amd/, "default" = the product build)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; TAG=$1; O=$R/gpurun_out/$TAG; mkdir -p $O
A=$2; B=$3; REPS=${4:-3}
X="--no-cpu-baseline --extras= --latency-steps 0"
for rep in $(seq 1 $REPS); do
  for v in A B; do
    lib=$A; [ $v = B ] && lib=$B
    if [ "$lib" = default ]; then unset GUBER_HIP_LIB; else export GUBER_HIP_LIB=$R/gubernator_amd/$lib; fi
    timeout 120 python bench.py $X > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
    python -c "import json; d=json.load(open('$O/bench_${v}_$rep.json')); print('$v', '$lib', round(d['value']/1e9,3), d['ms_per_step'], d.get('parity'), {k: v for k, v in d['roofline'].get('kernel_avg_us', {}).items() if 'multi' in k})"
  done
done
unset GUBER_HIP_LIB
