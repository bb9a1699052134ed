#!/bin/bash
# Round 5, call o: the analysis after its six slowest kernels were rewritten (same results: the device-analysis tests compare them with
# the host's) — tests, per-kernel profile, plan times, and the launch times the policy rests on.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05o; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_plan.log 2>&1; echo "pytest rc=$?" >> $O/pytest_plan.log
tail -5 $O/pytest_plan.log
grep -q "pytest rc=0" $O/pytest_plan.log || exit 1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/planprof -o p -- python scripts/plan_ms.py com-amazon-sbm --reps 10 > $O/plan_ms_profiled.log 2>&1
f=$(find /tmp/planprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/plan_kernel_stats.csv
GESPMM_PLAN_TIMING=1 timeout 300 python scripts/plan_ms.py com-amazon-sbm --reps 3 > $O/plan_timing.log 2>&1
timeout 900 python scripts/plan_ms.py com-amazon-sbm com-amazon-like pubmed-like geometric lfr-mu0.1 nws-k10 products-sbm --reps 5 2>&1 | grep -v amdgpu | cut -c1-400 > $O/plan_ms.log
timeout 900 python scripts/plan_ms.py com-amazon-sbm com-amazon-like geometric products-sbm --reps 5 --expected-launches 1000000 2>&1 | grep -v amdgpu | cut -c1-400 >> $O/plan_ms.log
head -24 $O/plan_kernel_stats.csv | cut -c1-160; grep -v amdgpu $O/plan_timing.log | tail -22; cut -c1-200 $O/plan_ms.log
