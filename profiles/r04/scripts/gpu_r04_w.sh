#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 1800 python scripts/holdout_audit.py --only lfr-verydense-mu0.2 lfr-verydense-mu0.5 --widths 32 64 128 256 512 2>&1 | grep -v "amdgpu.ids\|^W2026" > gpurun_out/r04/holdout_audit_dense.log
timeout 1800 python scripts/holdout_audit.py --standins --only reddit-sbm reddit-like --widths 64 128 256 2>&1 | grep -v "amdgpu.ids\|^W2026" >> gpurun_out/r04/holdout_audit_dense.log
cat gpurun_out/r04/holdout_audit_dense.log | cut -c1-420
