#!/bin/bash
# SQ counter passes (rocprofv3 --pmc, one group per pass, no tracing) over the headline configuration: what the three kernels of the
# owner-partitioned pipeline issue per wave, and how busy the issue ports are.   usage: gpu_r05_sq.sh <tag> [extra env, e.g. GUBER_FUSE_EP=0]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05_sq}; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --extras= --min-batches 256 --steps 256 --warmup 8 --profile-steps 0 --latency-steps 0"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" ; do
  i=$((i+1))
  rm -rf $O/p$i
  env "$@" timeout 200 rocprofv3 --pmc $grp --output-format csv -d $O/p$i -o pmc -- python $R/bench.py $ARGS > $O/p$i.log 2>&1
  echo "pass $i [$grp] rc=$?"; grep -i "error\|invalid\|not found" $O/p$i.log | head -3 | cut -c1-200
done
python - <<PY
import csv, glob, collections, json
out = collections.defaultdict(dict)
for path in sorted(glob.glob("$O/p*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0]
        if "guber::" in name:
            per[(name.replace("guber::", ""), row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (k, c), v in per.items():
        out[k][c] = (sum(v) / len(v), len(v))
with open("$O/sq_counters.txt", "w") as f:
    for k in sorted(out):
        if not any(s in k for s in ("k_part", "k_own", "k_eval3", "k_evalpart", "k_front", "k_eval2")): continue
        print(k, file=f); print(k)
        for c in sorted(out[k]):
            line = f"    {c:28s} {out[k][c][0]:16.1f}   (n={out[k][c][1]})"
            print(line, file=f); print(line)
        w = out[k].get("SQ_WAVES", (0, 0))[0]
        if w:
            line = "    per wave: " + "  ".join(f"{c[9:]} {out[k][c][0] / w:.0f}" for c in sorted(out[k]) if c.startswith("SQ_INSTS_"))
            print(line, file=f); print(line)
json.dump({k: {c: {"avg_per_launch": v[0], "launches": v[1]} for c, v in cs.items()} for k, cs in out.items()}, open("$O/sq_counters.json", "w"), indent=1)
PY
rm -rf $O/p*/ 2>/dev/null
