#!/bin/bash
# Round 5, call ag: soak of the staged kernels with 80 KB blocks at every width (N = 16 ... 1024), then the round's evidence run.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05ag; mkdir -p $O
timeout 1500 python scripts/staged_soak.py 70000 2000 2>&1 | grep -v amdgpu | tail -3 > $O/staged_soak_lds5_all_widths.log
cat $O/staged_soak_lds5_all_widths.log
grep -q "all equal" $O/staged_soak_lds5_all_widths.log || exit 1
bash scripts/gpu_profile_r05.sh
