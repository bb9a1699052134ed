#!/bin/bash
# Round 5, call s: does touching the staging list / tasks of the block D positions ahead (same XCD slice) shorten that block's first round
# trip? Experiment library (profiles/r05/experiments/staged_prefetch_build.py), D = GESPMM_STAGED_PF.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05s; mkdir -p $O
cp profiles/r05/experiments/_build/prefetch/libgespmm.so gespmm_amd/lib/libgespmm.so
for d in 0 16 32 64 128 256; do
  GESPMM_STAGED_PF=$d timeout 900 python scripts/kernel_ab.py --graphs com-amazon-sbm geometric nws-k10 lfr-mu0.1 products-sbm --widths 128 --kernels staged --tag "touch-ahead=$d " 2>&1 | grep -v amdgpu >> $O/staged_touch_ahead.log
done
cat $O/staged_touch_ahead.log
