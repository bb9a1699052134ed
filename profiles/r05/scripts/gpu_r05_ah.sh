#!/bin/bash
# Round 5, call ah: rows per block at 256-column tiles again with 80 staged rows per block (64 / 48 were chosen with 64).
set -x
export TMPDIR=/tmp
O=gpurun_out/r05ah; mkdir -p $O
for round in 1 2; do for r in 48 64 80 96; do
  GESPMM_STAGED_ROWS=$r timeout 900 python scripts/kernel_ab.py --graphs com-amazon-sbm geometric nws-k10 lfr-mu0.1 products-sbm --widths 256 --kernels staged --tag "round=$round rows=$r " 2>&1 | grep -v amdgpu >> $O/staged_rows_256_lds5.log
done; done
sort -k3,3 -k2,2 $O/staged_rows_256_lds5.log | cut -c1-120
