#!/bin/bash
# Round 5, call w: does a default-life plan (three levels, three sweeps) launch slower than a steady-state plan at wide N?
set -x
export TMPDIR=/tmp
O=gpurun_out/r05w; mkdir -p $O
timeout 2400 python scripts/plan_life_compare.py --graphs com-amazon-sbm com-amazon-like geometric lfr-mu0.1 nws-k10 products-sbm --widths 32 128 256 512 2>&1 | grep -v amdgpu > $O/plan_life_compare.log
cat $O/plan_life_compare.log
