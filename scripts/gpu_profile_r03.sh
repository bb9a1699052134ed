#!/bin/bash
# Round-3 evidence: (1) rocprofv3 kernel-trace stats + separate --pmc passes of the bench command for the four
# (stand-in, launch) pairs bench.py quotes traffic for; (2) kernel stats of the default bench run; (3) the analysis stage:
# timings, kernel breakdown, AUTO-plan audit, GCN epochs with / without plans; (4) SDDMM audit.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python bench.py --no-extra --no-cpu-baseline --steps 50 --warmup 5"
scripts/gpu_pmc.sh bench_sbm_plan "spmm_(seg)?stream" -- $B > gpurun_out/pmc_bench_sbm_plan.log 2>&1
scripts/gpu_pmc.sh bench_sbm_plain "spmm_(seg)?stream" -- $B --no-plan > gpurun_out/pmc_bench_sbm_plain.log 2>&1
scripts/gpu_pmc.sh bench_like_plan "spmm_(seg)?stream" -- $B --graph com-amazon-like > gpurun_out/pmc_bench_like_plan.log 2>&1
scripts/gpu_pmc.sh bench_like_plain "spmm_(seg)?stream" -- $B --graph com-amazon-like --no-plan > gpurun_out/pmc_bench_like_plain.log 2>&1
for t in bench_sbm_plan bench_sbm_plain bench_like_plan bench_like_plain; do echo "== $t"; cut -d, -f6- gpurun_out/pmc_$t/summary.csv; grep spmm_ gpurun_out/pmc_$t/kernel_stats.csv | cut -c1-200; done
P=/tmp/prof_bench; rm -rf $P; mkdir -p $P gpurun_out/r03
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o b -- python bench.py > gpurun_out/r03/bench_under_profiler.log 2>&1
f=$(find $P -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r03/bench_default_kernel_stats.csv
python bench.py > gpurun_out/r03/bench_round3.log 2> gpurun_out/r03/bench_round3.err
python scripts/plan_time.py com-amazon-like com-amazon-sbm products-sbm products-like --host-big 2>&1 | grep -v "amdgpu.ids\|^W2026" | cut -c1-300 > gpurun_out/r03/plan_time.log
bash scripts/gpu_profile_plan.sh com-amazon-sbm > /dev/null 2>&1; cp gpurun_out/prof_plan/kernel_stats_com-amazon-sbm.csv gpurun_out/r03/plan_device_kernel_stats.csv
python scripts/plan_knobs.py com-amazon-sbm 128 > gpurun_out/r03/cluster_knobs.log 2>&1
python scripts/plan_knobs.py com-amazon-like 128 >> gpurun_out/r03/cluster_knobs.log 2>&1
python scripts/plan_knobs.py products-sbm 128 >> gpurun_out/r03/cluster_knobs.log 2>&1
timeout 1500 python scripts/plan_audit.py > gpurun_out/r03/plan_audit.log 2>&1
bash profiles/r02/scripts/gcn_plans.sh > gpurun_out/r03/gcn_plans.log 2>&1
tail -3 gpurun_out/r03/plan_time.log; tail -5 gpurun_out/r03/plan_audit.log; tail -12 gpurun_out/r03/gcn_plans.log
