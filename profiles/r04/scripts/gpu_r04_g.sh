#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 1200 python profiles/r04/experiments/task_interleave.py 2>&1 | grep -v "amdgpu.ids\|^W2026" > gpurun_out/r04/task_interleave.log
cat gpurun_out/r04/task_interleave.log | cut -c1-300
