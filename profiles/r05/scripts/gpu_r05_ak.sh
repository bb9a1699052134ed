#!/bin/bash
# Round 5, call ak: batch-stream against segmented-stream on short rows (planted communities, mean degree 2 ... 8) — which one should a
# plan without staged tables launch?
set -x
export TMPDIR=/tmp
O=gpurun_out/r05ak; mkdir -p $O
timeout 1500 python scripts/staged_degree_sweep.py 2,3,4,5,6,8 2>&1 | grep -v amdgpu > $O/short_rows_batch_vs_segmented.log
cat $O/short_rows_batch_vs_segmented.log
