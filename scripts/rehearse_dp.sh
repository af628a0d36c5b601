#!/bin/bash
# Rehearsal of the N>1 bench path on a ONE-GPU box: two ranks share cuda:0 and exchange over gloo
# (RCCL refuses two ranks on one device).  Usage: scripts/rehearse_dp.sh [extra bench args]
set -e
cd "$(dirname "$0")/.."
export DFD_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 2 --steps 5 --warmup 2 --batch 64 --no-cpu-baseline "$@"
