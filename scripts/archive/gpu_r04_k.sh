#!/bin/bash
# k_own LDS / register budget variants under the fused headline (same box)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/${1:-r04_k}; mkdir -p $O
for v in "" _vk256 _vk256e2 _vk192e2; do
  GUBER_HIP_LIB=$R/gubernator_amd/libguber_hip$v.so timeout 600 python bench.py --no-cpu-baseline --extras uniform > $O/bench$v.json 2> $O/bench$v.err
  python - <<PY
import json
d=json.load(open("$O/bench$v.json"))
print("variant '$v': value", round(d["value"]/1e9,3), "uniform", round(d["uniform"]["value"]/1e9,3), d["roofline"]["kernel_avg_us"])
PY
done
