#!/bin/bash
# First GPU pass: parity tests, smoke, bench, rocprof stats, sweep.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
timeout 900 python scripts/sweep.py --quick --graphs com-amazon-like,com-amazon-like@0.9 --ncols 32,128,512 > gpurun_out/sweep_quick.log 2>&1
tail -5 gpurun_out/smoke.log gpurun_out/pytest_gpu.log gpurun_out/bench.log
