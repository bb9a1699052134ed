#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 1200 python profiles/r04/experiments/levels_staged.py 2>&1 | grep -v "amdgpu.ids\|^W2026" > gpurun_out/r04/levels_staged.log
cat gpurun_out/r04/levels_staged.log
