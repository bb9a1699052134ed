#!/bin/bash
# Round 5, call p: the two audits of the kernel rules again — their explicit candidates are steady-state plans now, like the AUTO plan
# they are compared with (they were default-life plans: three levels / three sweeps against six / five, a comparison of clusterings).
set -x
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
export GESPMM_HOLDOUT_DIR=profiles/r05/holdout
timeout 1800 python scripts/holdout_audit.py 2>&1 | grep -v "amdgpu.ids\|^W2026" > $O/holdout_audit.log
timeout 1800 python scripts/holdout_audit.py --standins --widths 32 64 128 256 512 2>&1 | grep -v "amdgpu.ids\|^W2026" > $O/standin_audit.log
grep "<--\|worst" $O/holdout_audit.log $O/standin_audit.log | cut -c1-300
