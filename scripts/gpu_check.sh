#!/bin/bash
# What the driver does at round end, in one gpurun call:
#   gpurun --timeout 3000 -- 'bash scripts/gpu_check.sh'
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 1800 python -m pytest tests -m gpu -x -q -rs > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
for f in smoke pytest_gpu bench; do tail -n 3 gpurun_out/$f.log | cut -c1-300; done
