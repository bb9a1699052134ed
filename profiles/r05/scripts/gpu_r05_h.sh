#!/bin/bash
# Round 5, call h: plan creation time after the allocation consolidation; 8 KB of LDS per wavefront (one 16-wavefront block per CU, twice the
# staged rows and block height) against the shipped 4 KB on the hit-bound graphs.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_plan.py tests/test_gpu_plan_staged.py tests/test_gpu_plan_device.py -x -q > $O/pytest_plan.log 2>&1; echo "rc=$?" >> $O/pytest_plan.log
tail -3 $O/pytest_plan.log
timeout 600 python scripts/plan_ms.py com-amazon-sbm com-amazon-like pubmed-like geometric products-sbm > $O/plan_ms.log 2>&1
timeout 600 python scripts/plan_ms.py --expected-launches 100000 com-amazon-sbm com-amazon-like pubmed-like >> $O/plan_ms.log 2>&1
for kb in 4 8; do
  GESPMM_STAGED_LDS_KB=$kb timeout 900 python scripts/kernel_ab.py --graphs products-sbm geometric nws-k10 com-amazon-sbm lfr-mu0.1 --widths 128 256 --kernels staged --tag "lds=$kb " >> $O/staged_lds.log 2>&1
done
GESPMM_STAGED_LDS_KB=8 GESPMM_STAGED_ROWS=96 timeout 900 python scripts/kernel_ab.py --graphs products-sbm geometric com-amazon-sbm lfr-mu0.1 --widths 128 --kernels staged --tag "lds=8 rows=96 " >> $O/staged_lds.log 2>&1
GESPMM_STAGED_LDS_KB=8 GESPMM_STAGED_ROWS=128 timeout 900 python scripts/kernel_ab.py --graphs products-sbm geometric com-amazon-sbm lfr-mu0.1 --widths 128 --kernels staged --tag "lds=8 rows=128 " >> $O/staged_lds.log 2>&1
grep -v amdgpu $O/plan_ms.log | cut -c1-330; grep -v amdgpu $O/staged_lds.log
