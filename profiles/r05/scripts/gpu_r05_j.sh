#!/bin/bash
# Round 5, call j: block shapes of the narrow-width staged kernel (rows per block, 8 vs 16 wavefronts) at N = 32 / 64.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05j; mkdir -p $O
for r in 512 640 768 1024; do
  GESPMM_STAGED_NARROW_ROWS=$r timeout 600 python scripts/kernel_ab.py --graphs com-amazon-sbm geometric nws-k10 products-sbm lfr-mu0.1 --widths 32 --kernels staged --tag "rows=$r " >> $O/narrow_shapes.log 2>&1
done
for r in 256 384 512; do
  GESPMM_STAGED_NARROW_ROWS=$r timeout 600 python scripts/kernel_ab.py --graphs com-amazon-sbm geometric nws-k10 products-sbm lfr-mu0.1 --widths 64 --kernels staged --tag "rows=$r " >> $O/narrow_shapes.log 2>&1
done
for r in 192 256 384; do
  GESPMM_STAGED_NARROW_WAVES=8 GESPMM_STAGED_NARROW_ROWS=$r timeout 600 python scripts/kernel_ab.py --graphs com-amazon-sbm geometric nws-k10 products-sbm lfr-mu0.1 --widths 32 --kernels staged --tag "waves=8 rows=$r " >> $O/narrow_shapes.log 2>&1
done
grep -v amdgpu $O/narrow_shapes.log
