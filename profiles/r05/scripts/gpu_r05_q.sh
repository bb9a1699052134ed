#!/bin/bash
# Round 5, call q: the persistent form of the staged-rows kernel (one workgroup per CU walks a run of blocks, the next block's rows are
# staged into a second LDS buffer while the current one is walked) against the product form. Experiment: GESPMM_STAGED_PERSIST.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05q; mkdir -p $O
for pz in 0 1 512 1024; do
  GESPMM_STAGED_PERSIST=$pz timeout 900 python scripts/kernel_ab.py --graphs com-amazon-sbm geometric nws-k10 lfr-mu0.1 products-sbm --widths 128 256 --kernels staged --tag "persist=$pz " 2>&1 | grep -v amdgpu >> $O/staged_persist.log
done
cat $O/staged_persist.log
