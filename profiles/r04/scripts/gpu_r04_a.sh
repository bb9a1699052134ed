#!/bin/bash
# Round 4, first GPU pass: (1) the default bench run (is the final line small and parsed?), (2) the com-amazon-like plan
# regression bisect (round 2's library beside the current one), (3) PMC passes of the headline graph at N = 32 and N = 512.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
( time python bench.py > gpurun_out/r04/bench_first.log 2> gpurun_out/r04/bench_first.err ) 2> gpurun_out/r04/bench_first.time
tail -1 gpurun_out/r04/bench_first.log | wc -c
python profiles/r04/experiments/like_regression.py com-amazon-like 2>&1 | grep -v "amdgpu.ids\|^W2026" > gpurun_out/r04/like_regression.log
B="python bench.py --no-extra --no-cpu-baseline --steps 50 --warmup 5"
scripts/gpu_pmc.sh bench_sbm_plan_N32 "spmm_(seg)?stream" -- $B --ncols 32 > gpurun_out/r04/pmc_N32.log 2>&1
scripts/gpu_pmc.sh bench_sbm_plan_N512 "spmm_(seg)?stream" -- $B --ncols 512 > gpurun_out/r04/pmc_N512.log 2>&1
for t in bench_sbm_plan_N32 bench_sbm_plan_N512; do echo "== $t"; cut -d, -f6- gpurun_out/pmc_$t/summary.csv; grep spmm_ gpurun_out/pmc_$t/kernel_stats.csv | cut -c1-200; done
cat gpurun_out/r04/like_regression.log | cut -c1-400
tail -1 gpurun_out/r04/bench_first.log | cut -c1-4100
cat gpurun_out/r04/bench_first.time
timeout 1500 python scripts/holdout_audit.py 2>&1 | grep -v "amdgpu.ids\|^W2026" > gpurun_out/r04/holdout_audit.log
cat gpurun_out/r04/holdout_audit.log | cut -c1-330
