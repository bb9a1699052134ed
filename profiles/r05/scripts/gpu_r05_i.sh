#!/bin/bash
# Round 5, call i: the lane-group form of the staged-rows kernel (N = 16 / 32 / 64): parity, then against the streaming kernels.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05i; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_plan_staged.py -x -q > $O/pytest_staged.log 2>&1; echo "rc=$?" >> $O/pytest_staged.log
tail -3 $O/pytest_staged.log
timeout 900 python scripts/kernel_ab.py --graphs com-amazon-sbm geometric lfr-mu0.1 nws-k10 com-amazon-like --widths 32 64 16 --kernels stream seg-stream staged --auto > $O/kernel_ab_narrow.log 2>&1
timeout 900 python scripts/kernel_ab.py --graphs products-sbm --widths 32 64 16 --kernels stream seg-stream staged --auto >> $O/kernel_ab_narrow.log 2>&1
for r in 256 512; do
  GESPMM_STAGED_NARROW_ROWS=$r timeout 600 python scripts/kernel_ab.py --graphs com-amazon-sbm geometric products-sbm --widths 32 --kernels staged --tag "rows=$r " >> $O/kernel_ab_narrow.log 2>&1
done
grep -v amdgpu $O/kernel_ab_narrow.log
