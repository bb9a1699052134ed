#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 2400 python profiles/r04/experiments/constants_resweep.py 2>&1 | grep -v "amdgpu.ids\|^W2026" > gpurun_out/r04/constants_resweep.log
cat gpurun_out/r04/constants_resweep.log | cut -c1-260
