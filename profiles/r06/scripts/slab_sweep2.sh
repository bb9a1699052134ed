#!/bin/bash
# three 48 KB blocks per CU (96 staged rows, 8 gathers per chunk) against two 80 KB blocks (160 rows, 16 gathers): rows per block x ranges
cd $GRAFT_REPO_ROOT
for KB in 5 3; do for R in 64 96; do for P in 8 10 12 16; do
  GESPMM_SLAB_LDS_KB=$KB GESPMM_SLAB_ROWS=$R GESPMM_SLABS=$P SLAB_ONLY=1 timeout 200 python profiles/r06/scripts/slab_plan_time.py reddit-sbm 2>&1 | grep "staged-slabs" | sed "s/^/lds_kb=$KB rows=$R /" | cut -c1-250
done; done; done
