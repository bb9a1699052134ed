#!/bin/bash
# Round 5, call k: the narrow-width staged kernel after the register diet (value broadcasts behind the wait, compare masks), default block
# shapes (rows = slots) and its AUTO rule: GPU suite, 600-seed soak incl. N = 16 / 32 / 64, AUTO against the explicit kernels.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05k; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 1200 python scripts/staged_soak.py 12000 600 > $O/staged_soak.log 2>&1; tail -2 $O/staged_soak.log
timeout 900 python scripts/kernel_ab.py --graphs com-amazon-sbm geometric lfr-mu0.1 nws-k10 com-amazon-like holme-kim-m5 --widths 32 64 --kernels stream seg-stream staged --auto > $O/kernel_ab_narrow.log 2>&1
timeout 900 python scripts/kernel_ab.py --graphs products-sbm --widths 16 32 64 --kernels stream seg-stream staged --auto >> $O/kernel_ab_narrow.log 2>&1
grep -v amdgpu $O/kernel_ab_narrow.log
