#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 900 python scripts/sweep.py --quick --graphs com-amazon-like,com-amazon-like@0.9 --ncols 128,32 --rounds 2 > gpurun_out/sweep4.log 2>&1
timeout 900 python scripts/ksweep.py > gpurun_out/ksweep3.log 2>&1
timeout 600 python bench.py > gpurun_out/bench.log 2>&1
tail -3 gpurun_out/pytest_gpu.log
