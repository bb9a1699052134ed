#!/bin/bash
# Round 5, call e: the record-stream staged-rows kernel with the store hazard fixed (two wait states): full GPU suite, a 1 500-seed soak
# (forced staged plans, N = 128 ... 1024, valued / unweighted, tune), AUTO on the stand-ins and hold-outs.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05e; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 1500 python scripts/staged_soak.py 7000 1500 > $O/staged_soak.log 2>&1; tail -2 $O/staged_soak.log
timeout 900 python scripts/kernel_ab.py --graphs geometric nws-k10 lfr-mu0.1 lfr-mu0.3 com-amazon-sbm com-amazon-like holme-kim-m5 ba-m6 --widths 128 256 --kernels stream seg-stream staged --auto > $O/kernel_ab.log 2>&1
timeout 600 python scripts/kernel_ab.py --graphs products-sbm --widths 128 256 512 --kernels seg-stream staged --auto > $O/kernel_ab_products.log 2>&1
cat $O/kernel_ab.log $O/kernel_ab_products.log | grep -v amdgpu.ids
