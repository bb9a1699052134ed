#!/bin/bash
# Round 6, after the non-temporal C stores: do the block height and the gathers per chunk of the staged-rows kernel still sit at their optimum?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06resweep; mkdir -p $O
for rep in 1 2; do
for u in 16 8; do
  for r in 0 80 96 128 144; do
    export GESPMM_STAGED_U=$u; if [ $r = 0 ]; then unset GESPMM_STAGED_ROWS; else export GESPMM_STAGED_ROWS=$r; fi
    timeout 300 python scripts/kernel_ab.py --graphs com-amazon-sbm products-sbm --widths 128 --kernels staged --tag "U=$u rows=$r " 2>&1 | grep -v amdgpu | cut -c1-160 >> $O/resweep.log
  done
done
done
sort -s -k3,3 $O/resweep.log
