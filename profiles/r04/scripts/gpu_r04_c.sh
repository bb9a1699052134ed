#!/bin/bash
# Round 4, third GPU pass: column-tiled staged kernel + hold-out thresholds: GPU tests, audits again, narrow-width geometry
# on the headline graph, 128-byte XCD-bound column tiles on the slab path.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -q -rs > gpurun_out/r04/pytest_gpu_c.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/pytest_gpu_c.log
tail -15 gpurun_out/r04/pytest_gpu_c.log | cut -c1-300
timeout 900 python profiles/r04/experiments/narrow_headline.py 2>&1 | grep -v "amdgpu.ids\|^W2026" > gpurun_out/r04/narrow_headline.log
timeout 900 python scripts/slab_sweep.py --groups 8,32 --slab-rows 12288,16384,24576,32768 --rpw 0,4,8 2>&1 | grep -v "amdgpu.ids\|^W2026" > gpurun_out/r04/slab_sweep_group8.log
timeout 1500 python scripts/holdout_audit.py 2>&1 | grep -v "amdgpu.ids\|^W2026" > gpurun_out/r04/holdout_audit_c.log
timeout 1500 python scripts/holdout_audit.py --standins --widths 32 64 128 256 512 2>&1 | grep -v "amdgpu.ids\|^W2026" > gpurun_out/r04/standin_audit_c.log
cat gpurun_out/r04/narrow_headline.log gpurun_out/r04/slab_sweep_group8.log | cut -c1-300
grep -B1 "<--" gpurun_out/r04/holdout_audit_c.log gpurun_out/r04/standin_audit_c.log | cut -c1-300; tail -1 gpurun_out/r04/holdout_audit_c.log gpurun_out/r04/standin_audit_c.log
