#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 1500 python scripts/plan_regression_ab.py profiles/r04/experiments/_build/libgespmm_r03.so 2>&1 | grep -v "amdgpu.ids\|^W2026" > gpurun_out/r04/regression_ab_vs_r03.log
cat gpurun_out/r04/regression_ab_vs_r03.log | cut -c1-330
