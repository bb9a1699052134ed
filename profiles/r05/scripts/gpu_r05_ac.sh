#!/bin/bash
# Round 5, call ac: 5 KB of LDS per wavefront — two 80 KB blocks fill the CU's 160 KB exactly; 160 / 80 staged rows instead of 128 / 64.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05ac; mkdir -p $O
for kb in 4 5; do
  GESPMM_STAGED_LDS_KB=$kb timeout 900 python scripts/kernel_ab.py --graphs com-amazon-sbm geometric nws-k10 lfr-mu0.1 products-sbm --widths 128 256 --kernels staged --tag "lds_kb=$kb " 2>&1 | grep -v amdgpu >> $O/staged_lds5.log
done
for r in 112 128 144; do
  GESPMM_STAGED_LDS_KB=5 GESPMM_STAGED_ROWS=$r timeout 900 python scripts/kernel_ab.py --graphs com-amazon-sbm geometric lfr-mu0.1 products-sbm --widths 128 --kernels staged --tag "lds_kb=5 rows=$r " 2>&1 | grep -v amdgpu >> $O/staged_lds5.log
done
cat $O/staged_lds5.log
