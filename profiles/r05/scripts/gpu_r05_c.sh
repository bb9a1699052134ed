#!/bin/bash
# Round 5, call c: the staged-rows kernel on the record stream (row-end records, packed multiply-adds, mask-bit branches): parity first,
# then the hold-out graphs and stand-ins against the streaming kernels, phase clocks and issue counters of the new walk.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05c; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_plan_staged.py tests/test_gpu_plan.py -x -q > $O/pytest_staged.log 2>&1; echo "rc=$?" >> $O/pytest_staged.log
tail -5 $O/pytest_staged.log
timeout 900 python scripts/kernel_ab.py --graphs geometric nws-k10 lfr-mu0.1 lfr-mu0.3 com-amazon-sbm com-amazon-like holme-kim-m5 --widths 128 256 --kernels stream seg-stream staged > $O/kernel_ab.log 2>&1
timeout 600 python scripts/kernel_ab.py --graphs products-sbm --widths 128 256 512 --kernels seg-stream staged > $O/kernel_ab_products.log 2>&1
timeout 600 python scripts/kernel_ab.py --graphs com-amazon-sbm --widths 512 1024 --kernels stream staged >> $O/kernel_ab.log 2>&1
timeout 600 bash scripts/gpu_sq_pmc.sh geometric_staged2 spmm_staged -- python scripts/kernel_pmc_case.py geometric 128 staged 3 > /dev/null 2>&1
cp gpurun_out/sq_geometric_staged2.log $O/
cp gespmm_amd/lib/libgespmm.so /tmp/libgespmm_release.so
cp gespmm_amd/lib_instr/libgespmm.so gespmm_amd/lib/libgespmm.so
GESPMM_STAGED_DEBUG=4 timeout 600 python scripts/kernel_ab.py --graphs geometric --widths 128 --kernels staged --tag "clk " > $O/staged_clocks.log 2>&1
GESPMM_STAGED_DEBUG=4 timeout 600 python scripts/kernel_ab.py --graphs products-sbm --widths 128 --kernels staged --tag "clk " >> $O/staged_clocks.log 2>&1
cp /tmp/libgespmm_release.so gespmm_amd/lib/libgespmm.so
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
cat $O/kernel_ab.log $O/kernel_ab_products.log | grep -v amdgpu.ids; grep -v "^+" $O/staged_clocks.log | grep -v amdgpu.ids | cut -c1-300; tail -4 $O/pytest_gpu.log
