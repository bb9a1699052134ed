#!/bin/bash
# What the driver runs at round end (one GPU) plus a one-rank RCCL run of the multi-GPU mode.
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time python bench.py ) > gpurun_out/r02_bench.log 2>&1
( time python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --graph rmat --rmat-scale ${1:-22} --steps 5 --warmup 2 ) > gpurun_out/r02_bench_rmat_torchrun1.log 2>&1
tail -3 gpurun_out/r02_bench.log | cut -c1-400
tail -3 gpurun_out/r02_bench_rmat_torchrun1.log | cut -c1-400
