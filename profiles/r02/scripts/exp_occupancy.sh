#!/bin/bash
# Occupancy limiter experiment: unused dynamic LDS per workgroup caps the wavefronts per CU of the batch-stream kernel.
cd $GRAFT_REPO_ROOT
for d in 0 16000 24000 36000 50000 76000; do
  echo "== GESPMM_DEBUG_DYN_LDS=$d"
  GESPMM_DEBUG_DYN_LDS=$d python scripts/plan_bench.py --only-plan --kernel stream --graphs com-amazon-sbm,com-amazon-like 2>&1 | grep -v amdgpu.ids | cut -c1-70
done
