#!/bin/bash
# Round 5, call g: full GPU suite + bench line after: 3 clustering levels for plans that expect < 2000 launches, yardstick = fastest copy.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05g; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.log
cp profiles/bench_extra_last.json $O/ 2>/dev/null
GESPMM_PLAN_TIMING=1 timeout 300 python scripts/kernel_pmc_case.py com-amazon-sbm 128 auto 2 > $O/plan_phases.log 2>&1
grep "^{" $O/bench.log | cut -c1-1200
