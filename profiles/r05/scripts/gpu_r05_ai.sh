#!/bin/bash
# Round 5, call ai: the mean-degree sweep again with the retuned blocks (80 KB, 112 rows for short rows): where does staged-rows start to pay?
set -x
export TMPDIR=/tmp
O=gpurun_out/r05ai; mkdir -p $O
timeout 1500 python scripts/staged_degree_sweep.py 3,4,5,6,8,12,24,48 2>&1 | grep -v amdgpu > $O/staged_degree_sweep_retuned.log
cat $O/staged_degree_sweep_retuned.log
