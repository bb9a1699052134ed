#!/bin/bash
# End-of-round soak on the final sources: every launch form on random matrices, the device analysis against the host form, the staged tiles + tune.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
{
echo "# scripts/soak_fuzz.py --cases 2000 --seed 404"; timeout 1500 python scripts/soak_fuzz.py --cases 2000 --seed 404 2>&1 | grep -v "amdgpu.ids\|^W2026" | tail -4
echo "# scripts/plan_device_fuzz.py 400 404"; timeout 1500 python scripts/plan_device_fuzz.py 400 404 2>&1 | grep -v "amdgpu.ids\|^W2026" | tail -4
echo "# scripts/staged_soak.py 9000 1000"; timeout 1500 python scripts/staged_soak.py 9000 1000 2>&1 | grep -v "amdgpu.ids\|^W2026" | tail -3
} > gpurun_out/r04/soak_and_fuzz.log 2>&1
cat gpurun_out/r04/soak_and_fuzz.log | cut -c1-300
