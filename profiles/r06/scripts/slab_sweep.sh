#!/bin/bash
# rows per block x column ranges of the plan's slab tables on the dense community graph (one process per point: both knobs are read once)
cd $GRAFT_REPO_ROOT
for R in 64 96 128 160 192; do for P in 4 6 8 10 12 16; do
  GESPMM_SLAB_ROWS=$R GESPMM_SLABS=$P SLAB_ONLY=1 timeout 200 python profiles/r06/scripts/slab_plan_time.py reddit-sbm 2>&1 | grep "staged-slabs" | sed "s/^/rows=$R /" | cut -c1-260
done; done
