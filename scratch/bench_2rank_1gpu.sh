#!/bin/bash
# Exercise bench.py's multi-rank code path with 2 processes that share the one GPU of the box (gloo transport,
# CUDA tensors staged through the host by genjax_amd.distributed when the backend is gloo).
cd $GRAFT_REPO_ROOT
export GJX_DIST_BACKEND=gloo GJX_ALL_ON_DEVICE0=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 3 --k-per-gpu 262144 2>&1 | tail -5
