#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 1500 python profiles/r04/experiments/narrow_vec_rule.py 2>&1 | grep -v "amdgpu.ids\|^W2026" > gpurun_out/r04/narrow_vec_rule.log
timeout 600 python -m pytest tests/test_gpu_plan.py tests/test_gpu_plan_staged.py -m gpu -q -x > gpurun_out/r04/pytest_gpu_d.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/pytest_gpu_d.log
timeout 1500 python scripts/holdout_audit.py --widths 128 256 2>&1 | grep -v "amdgpu.ids\|^W2026" > gpurun_out/r04/holdout_audit_d.log
cat gpurun_out/r04/narrow_vec_rule.log | cut -c1-300
tail -3 gpurun_out/r04/pytest_gpu_d.log
grep "N=\|==" gpurun_out/r04/holdout_audit_d.log | cut -c1-200
