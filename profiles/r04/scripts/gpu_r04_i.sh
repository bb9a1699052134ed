#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_plan_staged.py -m gpu -q -x > gpurun_out/r04/pytest_gpu_i.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/pytest_gpu_i.log
tail -4 gpurun_out/r04/pytest_gpu_i.log | cut -c1-300
( timeout 2400 python scripts/staged_soak.py 7000 1500 2>&1 | grep -v "amdgpu.ids\|^W2026" | tail -5 ) > gpurun_out/r04/staged_soak.log
cat gpurun_out/r04/staged_soak.log | cut -c1-400
