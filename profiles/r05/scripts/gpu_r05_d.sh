#!/bin/bash
# Round 5, call d: 256-column tiles of the staged-rows kernel with ONE 16-byte store per lane at a row end (accumulators pinned to a
# register quadruple): parity, then N = 256 / 512 / 1024 against the streaming kernels.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05d; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_plan_staged.py -x -q > $O/pytest_staged.log 2>&1; echo "rc=$?" >> $O/pytest_staged.log
tail -3 $O/pytest_staged.log
timeout 900 python scripts/kernel_ab.py --graphs geometric nws-k10 lfr-mu0.1 com-amazon-sbm --widths 256 --kernels stream seg-stream staged > $O/kernel_ab.log 2>&1
timeout 600 python scripts/kernel_ab.py --graphs com-amazon-sbm --widths 128 512 1024 --kernels stream staged >> $O/kernel_ab.log 2>&1
timeout 600 python scripts/kernel_ab.py --graphs products-sbm --widths 256 512 --kernels seg-stream staged > $O/kernel_ab_products.log 2>&1
timeout 900 python scripts/staged_degree_sweep.py 4,6,8,12 > $O/staged_degree_sweep.log 2>&1
cat $O/kernel_ab.log $O/kernel_ab_products.log $O/staged_degree_sweep.log | grep -v amdgpu.ids
