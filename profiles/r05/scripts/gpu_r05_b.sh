#!/bin/bash
# Round 5, call b: the copy yardstick's launch shapes, phase clocks of the staged-rows kernel (instrumented build), where the staged kernel
# starts to pay on short rows (degree sweep), issue-side counters of the staged kernel on the geometric hold-out graph and the headline.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05b; mkdir -p $O
timeout 300 python scripts/copy_yardstick.py > $O/copy_yardstick.log 2>&1
timeout 900 python scripts/staged_degree_sweep.py 4,6,8,12 > $O/staged_degree_sweep.log 2>&1
timeout 600 bash scripts/gpu_sq_pmc.sh geometric_staged spmm_staged -- python scripts/kernel_pmc_case.py geometric 128 staged 3 > /dev/null 2>&1
cp gpurun_out/sq_geometric_staged.log $O/
timeout 600 bash scripts/gpu_sq_pmc.sh sbm_staged spmm_staged -- python scripts/kernel_pmc_case.py com-amazon-sbm 128 staged 3 > /dev/null 2>&1
cp gpurun_out/sq_sbm_staged.log $O/
timeout 600 bash scripts/gpu_sq_pmc.sh lfr01_stream spmm_stream -- python scripts/kernel_pmc_case.py lfr-mu0.1 128 stream 3 > /dev/null 2>&1
cp gpurun_out/sq_lfr01_stream.log $O/
cp gespmm_amd/lib/libgespmm.so /tmp/libgespmm_release.so
cp gespmm_amd/lib_instr/libgespmm.so gespmm_amd/lib/libgespmm.so
GESPMM_STAGED_DEBUG=4 timeout 600 python scripts/kernel_ab.py --graphs geometric com-amazon-sbm nws-k10 lfr-mu0.1 --widths 128 --kernels staged --tag "clk " > $O/staged_clocks.log 2>&1
GESPMM_STAGED_DEBUG=4 timeout 600 python scripts/kernel_ab.py --graphs products-sbm --widths 128 --kernels staged --tag "clk " >> $O/staged_clocks.log 2>&1
cp /tmp/libgespmm_release.so gespmm_amd/lib/libgespmm.so
cat $O/copy_yardstick.log $O/staged_degree_sweep.log; grep -v "^+" $O/staged_clocks.log | cut -c1-300
