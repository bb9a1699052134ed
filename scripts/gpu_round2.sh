#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
timeout 1200 python scripts/sweep.py --graphs com-amazon-like,com-amazon-like@0.9 --ncols 128,32,512 --rounds 2 > gpurun_out/sweep2.log 2>&1
timeout 900 python scripts/sweep.py --quick --graphs reddit-like --ncols 128 --rounds 2 --out gpurun_out/sweep_reddit.json > gpurun_out/sweep_reddit.log 2>&1
tail -3 gpurun_out/pytest_gpu.log gpurun_out/bench.log
