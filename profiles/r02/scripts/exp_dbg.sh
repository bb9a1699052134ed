#!/bin/bash
cd $GRAFT_REPO_ROOT
for d in 0 1 2 3 4; do
  echo "== GESPMM_LDSROW_DEBUG=$d"
  GESPMM_LDSROW_DEBUG=$d python scripts/plan_bench.py --only-plan --kernel lds-rows --graphs com-amazon-sbm,com-amazon-like 2>&1 | grep -v amdgpu.ids | cut -c1-90
done
