#!/bin/bash
# Round 4, second GPU pass: the refactored library (plan_policy, 6 clustering levels): GPU tests, hold-out audit (all graphs),
# the same audit on the repository's stand-ins, clustering-level sweep.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 1800 python -m pytest tests -m gpu -x -q -rs > gpurun_out/r04/pytest_gpu_b.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/pytest_gpu_b.log
tail -5 gpurun_out/r04/pytest_gpu_b.log
timeout 1500 python scripts/holdout_audit.py 2>&1 | grep -v "amdgpu.ids\|^W2026" > gpurun_out/r04/holdout_audit_b.log
timeout 1500 python scripts/holdout_audit.py --standins --widths 32 64 128 256 512 2>&1 | grep -v "amdgpu.ids\|^W2026" > gpurun_out/r04/standin_audit_b.log
cat gpurun_out/r04/holdout_audit_b.log gpurun_out/r04/standin_audit_b.log | cut -c1-360
