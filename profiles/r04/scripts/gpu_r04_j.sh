#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 1500 python profiles/r04/experiments/staged_n64.py 2>&1 | grep -v "amdgpu.ids\|^W2026" > gpurun_out/r04/staged_n64.log
cat gpurun_out/r04/staged_n64.log | cut -c1-300
timeout 300 python -m pytest tests/test_gpu_plan.py -m gpu -q -x -k tune 2>&1 | tail -3
