#!/bin/bash
# Round 5, call l: block counts that fill whole generations of block slots (headline graph, N = 32), 16 gathers per chunk at N = 128.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05l; mkdir -p $O
for r in 164 219 328 512; do
  GESPMM_STAGED_NARROW_ROWS=$r timeout 600 python scripts/kernel_ab.py --graphs com-amazon-sbm --widths 32 64 --kernels staged --tag "rows=$r " >> $O/narrow_generations.log 2>&1
done
for u in 8 16; do
  GESPMM_STAGED_U=$u timeout 900 python scripts/kernel_ab.py --graphs products-sbm geometric nws-k10 com-amazon-sbm lfr-mu0.1 --widths 128 --kernels staged --tag "U=$u " >> $O/staged_u.log 2>&1
done
grep -v amdgpu $O/narrow_generations.log $O/staged_u.log
