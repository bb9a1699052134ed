#!/bin/bash
# Round 5, call ad: 80 KB blocks (5 KB of LDS per wavefront) as the default of the staged-rows kernel: GPU suite, soak, every width.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05ad; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
grep -q "pytest rc=0" $O/pytest.log || exit 1
timeout 1500 python scripts/staged_soak.py 60000 1500 2>&1 | grep -v amdgpu | tail -3 > $O/staged_soak_lds5.log
cat $O/staged_soak_lds5.log
for kb in 4 5; do
  GESPMM_STAGED_LDS_KB=$kb timeout 1500 python scripts/kernel_ab.py --graphs com-amazon-sbm geometric nws-k10 products-sbm --widths 128 256 512 1024 --kernels staged --tag "lds_kb=$kb " 2>&1 | grep -v amdgpu >> $O/staged_lds5_widths.log
done
cat $O/staged_lds5_widths.log
