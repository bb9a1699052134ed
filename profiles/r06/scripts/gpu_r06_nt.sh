#!/bin/bash
# Round 6: `sc1 nt` C stores in the staged kernels (product) — parity tests, the streaming kernels' store policies, a bench line.
export TMPDIR=/tmp
O=gpurun_out/r06nt; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_plan_staged.py tests/test_gpu_plan_staged_general.py tests/test_gpu_plan.py tests/test_gpu_plan_device.py -x -q > $O/pytest_staged.log 2>&1; echo "rc=$?" >> $O/pytest_staged.log
tail -3 $O/pytest_staged.log
timeout 1200 python profiles/r06/scripts/store_flags_sweep.py 2>&1 | grep -v amdgpu > $O/store_flags_sweep.log
cat $O/store_flags_sweep.log
timeout 600 python bench.py --no-extra 2>/dev/null | tail -1 > $O/bench_noextra.json
cut -c1-600 $O/bench_noextra.json
