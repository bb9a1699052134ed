#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 2400 python scripts/holdout_audit.py --widths 16 64 512 2>&1 | grep -v "amdgpu.ids\|^W2026" > gpurun_out/r04/holdout_audit_other_widths.log
grep "<--\|worst\|==" gpurun_out/r04/holdout_audit_other_widths.log | cut -c1-300
