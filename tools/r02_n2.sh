#!/bin/bash
# rehearsal of bench.py's N = 2 code path on the one-GPU box: 2 ranks over gloo sharing cuda:0 (RCCL refuses two ranks on one device)
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 20 --warmup 6 --backend gloo > $R/gpurun_out/n2.out 2> $R/gpurun_out/n2.err
echo "rc=$?"; cut -c1-600 $R/gpurun_out/n2.out; tail -3 $R/gpurun_out/n2.err | cut -c1-300
