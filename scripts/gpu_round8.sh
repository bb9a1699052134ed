#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 900 python scripts/exp2.py > gpurun_out/exp2.log 2>&1
timeout 900 python scripts/exp1.py > gpurun_out/exp1.log 2>&1
timeout 600 python bench.py > gpurun_out/bench.log 2>&1
tail -3 gpurun_out/pytest_gpu.log
