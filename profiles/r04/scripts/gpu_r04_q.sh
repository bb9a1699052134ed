#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 900 python profiles/r04/experiments/far_marks_headline.py 2>&1 | grep -v "amdgpu.ids\|^W2026" > gpurun_out/r04/far_marks_headline.log
cat gpurun_out/r04/far_marks_headline.log
