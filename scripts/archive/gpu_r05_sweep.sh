#!/bin/bash
# PREPARED FOR ROUND 5 (round 4 ran out of GPU time): batches in flight on the round-4 kernels — shards x streams (a launch carries the
# next batch of up to GUBER_MULTI_MAX = 4 shards of a stream; more shards than streams x 4 = several launch groups per stream), and the
# owner count pinned either way beside the traffic-following default.  One box; every run under its own short timeout.
#   usage: gpu_r05_sweep.sh [tag]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=$R/gpurun_out/${1:-r05_sweep}; mkdir -p $O
run() {  # name, bench args
  local name=$1; shift
  timeout 60 python bench.py --no-cpu-baseline --extras "" --latency-steps 0 "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?"
}
# FIRST: GUBER_FUSE_EP (k_eval3 of a group + k_part of the same tables' next group in ONE launch: two launches per pass instead of three;
# built at the end of round 4, checked through the kernel source on the CPU only) — parity on the GPU, then the headline with / without,
# alternating on this box; the bench's own parity gate (2048 / 2048 batches by digest) runs in both
GUBER_FUSE_EP=1 timeout 300 python tests/fuse_ep_check.py > $O/fuse_ep_check.txt 2>&1; echo "fuse_ep check rc=$?"; tail -4 $O/fuse_ep_check.txt
if grep -q "FUSE_EP CHECK OK" $O/fuse_ep_check.txt; then
  for rep in 1 2; do
    for ep in 0 1; do
      GUBER_FUSE_EP=$ep timeout 90 python bench.py --no-cpu-baseline --extras "" --latency-steps 0 > $O/bench_ep${ep}_$rep.json 2> $O/bench_ep${ep}_$rep.err; echo "bench fuse_ep=$ep rep $rep rc=$?"
      python -c "import json; d=json.load(open('$O/bench_ep${ep}_$rep.json')); print('fuse_ep=$ep', round(d['value']/1e9,3), d['ms_per_step'], d.get('parity'), {k: v for k, v in d['roofline'].get('kernel_avg_us', {}).items() if 'multi' in k})"
    done
  done
  for cfg in "12 2" "16 4" "24 4" "24 3"; do                 # (more than four shards per stream: several groups per stream and round, each with its own held-back k_eval3)
    set -- $cfg
    GUBER_FUSE_EP=1 timeout 90 python bench.py --no-cpu-baseline --extras "" --latency-steps 0 --shards $1 --streams $2 > $O/bench_ep1_s$1_t$2.json 2> $O/bench_ep1_s$1_t$2.err
    python -c "import json; d=json.load(open('$O/bench_ep1_s$1_t$2.json')); print('fuse_ep=1 $1 shards $2 streams', round(d['value']/1e9,3), d['ms_per_step'])"
  done
fi
NAMES=""
for cfg in "12 3" "12 4" "16 4" "16 3" "18 3" "20 5" "24 4" "24 6" "8 2" "12 2"; do
  set -- $cfg
  run s$1_t$2 --shards $1 --streams $2; NAMES="$NAMES s$1_t$2"
done
python - <<PY
import json
for f in "$NAMES".split():
    try: d = json.load(open("$O/bench_%s.json" % f))
    except Exception as e: print(f, "unreadable", e); continue
    print(f, "value", round(d["value"]/1e9, 3), "ms/step", d["ms_per_step"], {k: v for k, v in d["roofline"].get("kernel_avg_us", {}).items() if "multi" in k})
PY
# six tables per launch instead of four (a build: make -C gubernator_amd/csrc variant VNAME=mm6 VFLAGS=-DGUBER_MULTI_MAX=6, BEFORE the call)
if [ -f $R/gubernator_amd/libguber_hip_v_mm6.so ]; then
  for cfg in "18 3" "12 2" "24 4" "12 3"; do
    set -- $cfg
    GUBER_HIP_LIB=$R/gubernator_amd/libguber_hip_v_mm6.so timeout 60 python bench.py --no-cpu-baseline --extras "" --latency-steps 0 --shards $1 --streams $2 > $O/bench_mm6_s$1_t$2.json 2> $O/bench_mm6_s$1_t$2.err
    python -c "import json; d=json.load(open('$O/bench_mm6_s$1_t$2.json')); print('mm6 s$1 t$2', round(d['value']/1e9,3), d['ms_per_step'])"
  done
fi
# the device wire decoder's table walk against the serial one: parity (against the host transcoder) and the rates by RPC size
for t in 0 1; do
  GUBER_WIRE_TABLE=$t timeout 120 python -m pytest tests/test_gpu_wire_dev.py -m gpu -q -s > $O/pytest_wire_table$t.txt 2>&1; echo "wire decode, GUBER_WIRE_TABLE=$t rc=$?"
  grep -E "device wire decode|passed|failed" $O/pytest_wire_table$t.txt | cut -c1-200
done
