#!/bin/bash
# Round 5, call aj: short rows keep their staged tables from a share of 0.42 at N = 128 (mean degree 4-8): the degree sweep again (what does
# AUTO take now), then the round's evidence run on the final sources.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05aj; mkdir -p $O
timeout 1500 python scripts/staged_degree_sweep.py 3,4,5,6,8 2>&1 | grep -v amdgpu > $O/staged_degree_sweep_after_rule.log
cat $O/staged_degree_sweep_after_rule.log
bash scripts/gpu_profile_r05.sh
