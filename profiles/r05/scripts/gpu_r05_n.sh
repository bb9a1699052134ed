#!/bin/bash
# Round 5, call n: where do the narrow staged kernel's microseconds go on short rows at N = 32?  Ablation libraries (built by
# profiles/r05/experiments/narrow_ablate_build.py: no memory gathers / no stores / no staging copy / all three / no walk) and the
# issue-unit counters of the real kernel. (The ablated libraries compute WRONG products on purpose: only their times are read.)
set -x
export TMPDIR=/tmp
O=gpurun_out/r05n; mkdir -p $O
cp gespmm_amd/lib/libgespmm.so /tmp/libgespmm_product.so
for c in product nomem nostore nostage all nowalk; do
  if [ $c = product ]; then cp /tmp/libgespmm_product.so gespmm_amd/lib/libgespmm.so; else cp profiles/r05/experiments/_build/$c/libgespmm.so gespmm_amd/lib/libgespmm.so; fi
  timeout 600 python scripts/kernel_ab.py --graphs com-amazon-sbm geometric --widths 32 64 --kernels staged --tag "$c " >> $O/narrow_ablation.log 2>&1
done
cp /tmp/libgespmm_product.so gespmm_amd/lib/libgespmm.so
bash scripts/gpu_sq_pmc.sh narrow_sbm_N32 spmm_staged_narrow -- python scripts/kernel_pmc_case.py com-amazon-sbm 32 staged 5 > /dev/null 2>&1
cp gpurun_out/sq_narrow_sbm_N32.log $O/
bash scripts/gpu_sq_pmc.sh stream_sbm_N32 spmm_stream -- python scripts/kernel_pmc_case.py com-amazon-sbm 32 stream 5 > /dev/null 2>&1
cp gpurun_out/sq_stream_sbm_N32.log $O/
GESPMM_PLAN_TIMING=1 timeout 300 python scripts/plan_ms.py com-amazon-sbm --reps 3 > $O/plan_timing.log 2>&1
for sw in 2 3 4 5; do
  GESPMM_CLUSTER_SWEEPS=$sw timeout 600 python scripts/plan_ms.py com-amazon-sbm geometric lfr-mu0.1 --reps 3 2>&1 | grep -v amdgpu | cut -c1-330 | sed "s/^/sweeps=$sw /" >> $O/cluster_sweeps.log
  GESPMM_CLUSTER_SWEEPS=$sw timeout 600 python scripts/kernel_ab.py --graphs com-amazon-sbm geometric lfr-mu0.1 nws-k10 products-sbm --widths 128 --kernels staged --tag "sweeps=$sw " 2>&1 | grep -v amdgpu >> $O/cluster_sweeps.log
done
cat $O/cluster_sweeps.log
grep -v amdgpu $O/narrow_ablation.log; tail -60 $O/plan_timing.log; cat $O/sq_narrow_sbm_N32.log $O/sq_stream_sbm_N32.log
