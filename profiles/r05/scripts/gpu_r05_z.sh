#!/bin/bash
# Round 5, call z: what the shallow plans lose on Holme-Kim (7-9 %) — levels or sweeps? (levels, sweeps) through GESPMM_CLUSTER_*.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05z; mkdir -p $O
for lv in 3 6; do for sw in 3 5; do
  GESPMM_CLUSTER_LEVELS=$lv GESPMM_CLUSTER_SWEEPS=$sw timeout 900 python scripts/plan_life_compare.py --graphs holme-kim-m5 ba-m6 lfr-mu0.5 com-amazon-sbm geometric nws-k10 --widths 128 --lives 200 2>&1 | grep -v amdgpu | sed "s/^/levels=$lv sweeps=$sw /" >> $O/levels_vs_sweeps.log
done; done
cat $O/levels_vs_sweeps.log
