#!/bin/bash
# Round-2 evidence for bench.py's roofline block: rocprofv3 kernel-trace stats + separate --pmc passes of the bench command
# (headline through the plan, headline through the plain call, the community stand-in through the plan).
cd $GRAFT_REPO_ROOT
B="python bench.py --no-extra --no-cpu-baseline --steps 50 --warmup 5"
scripts/gpu_pmc.sh bench_plan "spmm_(seg)?stream" -- $B > gpurun_out/pmc_bench_plan.log 2>&1
scripts/gpu_pmc.sh bench_plain "spmm_(seg)?stream" -- $B --no-plan > gpurun_out/pmc_bench_plain.log 2>&1
scripts/gpu_pmc.sh bench_sbm_plan "spmm_(seg)?stream" -- $B --graph com-amazon-sbm > gpurun_out/pmc_bench_sbm_plan.log 2>&1
for t in bench_plan bench_plain bench_sbm_plan; do echo "== $t"; cut -d, -f6- gpurun_out/pmc_$t/summary.csv; grep spmm_ gpurun_out/pmc_$t/kernel_stats.csv | cut -c1-200; done
