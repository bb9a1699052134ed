#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 900 python profiles/r04/experiments/small_graph_plans.py 2>&1 | grep -v "amdgpu.ids\|^W2026" > gpurun_out/r04/small_graph_plans.log
cat gpurun_out/r04/small_graph_plans.log | cut -c1-330
timeout 300 python -m pytest tests/test_gpu_op.py -m gpu -q -x 2>&1 | tail -2
