#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 900 python scripts/exp3.py > gpurun_out/exp3.log 2>&1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 50 --warmup 5 > gpurun_out/bench_torchrun1.log 2>&1
timeout 900 python bench.py --graph rmat --rmat-scale 24 --ncols 256 --steps 10 --warmup 2 --no-extra --no-cpu-baseline > gpurun_out/bench_rmat24.log 2>&1
timeout 600 python examples/gcn_custom.py --dataset reddit-like --n-hidden 128 --epochs 20 > gpurun_out/gcn_reddit.log 2>&1
tail -3 gpurun_out/pytest_gpu.log
