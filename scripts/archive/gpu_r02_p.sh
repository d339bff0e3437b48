#!/bin/bash
# host link ceilings + the N>1 launch line of the driver on one device (gloo, two ranks sharing GPU 0)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r02_p; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/pcie_peak tools/pcie_peak.hip 2>/dev/null && timeout 300 /tmp/pcie_peak | tee $O/pcie_peak.txt
echo "== torchrun 2 ranks on one device (gloo)"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 16 --warmup 4 --keys 2000000 --one-device --backend gloo --no-cpu-baseline --extras "" > $O/torchrun2.json 2> $O/torchrun2.err; echo "rc=$?"; tail -c 1500 $O/torchrun2.json; tail -5 $O/torchrun2.err | cut -c1-300
