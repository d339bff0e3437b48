#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r02_p2; mkdir -p $O
echo "== torchrun 2 ranks on one device (gloo), default dispatch"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 16 --warmup 4 --keys 2000000 --one-device --backend gloo --no-cpu-baseline --extras "" > $O/torchrun2.json 2> $O/torchrun2.err; echo "rc=$?"; python -c "
import json; d=json.loads(open('$O/torchrun2.json').read().strip().splitlines()[-1]); print(d['value'], d['n_gpus'], d['config']['dispatch'], d['roofline']['kernel'], d['roofline']['frac'])"; tail -3 $O/torchrun2.err | cut -c1-300
