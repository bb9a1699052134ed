#!/bin/bash
# Round 5, call aa: rows per block of the staged-rows kernel again, now that the walk is the record stream (96 / 64 came from round 3's walk).
set -x
export TMPDIR=/tmp
O=gpurun_out/r05aa; mkdir -p $O
for r in 64 80 96 112 128 160; do
  GESPMM_STAGED_ROWS=$r timeout 900 python scripts/kernel_ab.py --graphs com-amazon-sbm geometric nws-k10 lfr-mu0.1 products-sbm --widths 128 --kernels staged --tag "rows=$r " 2>&1 | grep -v amdgpu >> $O/staged_rows_per_block.log
done
for r in 32 48 64 80 96; do
  GESPMM_STAGED_ROWS=$r timeout 900 python scripts/kernel_ab.py --graphs com-amazon-sbm geometric products-sbm --widths 256 --kernels staged --tag "rows=$r " 2>&1 | grep -v amdgpu >> $O/staged_rows_per_block.log
done
cat $O/staged_rows_per_block.log
