#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
python profiles/r04/experiments/plan_phase_times.py 2>&1 | grep -v "amdgpu.ids\|^W2026" > gpurun_out/r04/plan_phase_times.log
grep -A40 "creation 2" gpurun_out/r04/plan_phase_times.log | cut -c1-100
