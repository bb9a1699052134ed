#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_plan_staged.py tests/test_gpu_driver.py -m gpu -q -x 2>&1 | tail -3
timeout 1800 python scripts/holdout_audit.py --widths 128 256 512 2>&1 | grep -v "amdgpu.ids\|^W2026" > gpurun_out/r04/holdout_audit_far_off.log
timeout 1800 python scripts/holdout_audit.py --standins --widths 128 256 512 2>&1 | grep -v "amdgpu.ids\|^W2026" > gpurun_out/r04/standin_audit_far_off.log
grep "<--\|worst" gpurun_out/r04/holdout_audit_far_off.log gpurun_out/r04/standin_audit_far_off.log | cut -c1-300
grep -A1 "N=" gpurun_out/r04/holdout_audit_far_off.log gpurun_out/r04/standin_audit_far_off.log | grep "staged" | cut -c1-250
