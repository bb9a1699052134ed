#!/bin/bash
# Round 5, call ab: rows per staged block by mean degree (112 / 96 at 128 columns, 64 / 48 at 256-column tiles): GPU suite, the kernels
# side by side, both audits.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05ab; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 900 python scripts/kernel_ab.py --graphs com-amazon-sbm geometric nws-k10 lfr-mu0.1 lfr-mu0.3 products-sbm --widths 128 256 512 --kernels stream staged --auto 2>&1 | grep -v amdgpu > $O/kernel_ab_rows_rule.log
cat $O/kernel_ab_rows_rule.log
export GESPMM_HOLDOUT_DIR=profiles/r05/holdout
timeout 1800 python scripts/holdout_audit.py 2>&1 | grep -v "amdgpu.ids\|^W2026" > $O/holdout_audit.log
timeout 1800 python scripts/holdout_audit.py --standins --widths 32 64 128 256 512 2>&1 | grep -v "amdgpu.ids\|^W2026" > $O/standin_audit.log
grep "<--\|worst" $O/holdout_audit.log $O/standin_audit.log | cut -c1-300
