#!/bin/bash
# Round 5, call r: soak of the staged-rows plans after the analysis kernels were reworked (staging lists, transposition, model):
# random graphs and widths, every product compared bit for bit with the plain call's.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05r; mkdir -p $O
timeout 2400 python scripts/staged_soak.py 40000 2500 2>&1 | grep -v amdgpu | tail -25 > $O/staged_soak_after_analysis_rework.log
tail -5 $O/staged_soak_after_analysis_rework.log
