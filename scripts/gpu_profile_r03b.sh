#!/bin/bash
# Round 3, after the staged-rows kernel: (1) the four PMC captures bench.py quotes traffic for, again (the stamps follow the
# sources); (2) counters of the products-shaped community graph at N = 128 through the staged kernel and through the
# segmented-stream kernel of the same plan; (3) the default bench run.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python bench.py --no-extra --no-cpu-baseline --steps 50 --warmup 5"
scripts/gpu_pmc.sh bench_sbm_plan "spmm_(seg)?stream" -- $B > gpurun_out/pmc_bench_sbm_plan.log 2>&1
scripts/gpu_pmc.sh bench_sbm_plain "spmm_(seg)?stream" -- $B --no-plan > gpurun_out/pmc_bench_sbm_plain.log 2>&1
scripts/gpu_pmc.sh bench_like_plan "spmm_(seg)?stream" -- $B --graph com-amazon-like > gpurun_out/pmc_bench_like_plan.log 2>&1
scripts/gpu_pmc.sh bench_like_plain "spmm_(seg)?stream" -- $B --graph com-amazon-like --no-plan > gpurun_out/pmc_bench_like_plain.log 2>&1
scripts/gpu_pmc.sh products_sbm_staged "spmm_staged" -- python profiles/r03/experiments/narrow_rows_sbm.py 128 > gpurun_out/pmc_products_sbm_staged.log 2>&1
scripts/gpu_pmc.sh products_sbm_seg "spmm_segstream" -- python profiles/r03/experiments/narrow_rows_sbm.py 128 seg-stream > gpurun_out/pmc_products_sbm_seg.log 2>&1
for t in bench_sbm_plan bench_sbm_plain bench_like_plan bench_like_plain products_sbm_staged products_sbm_seg; do echo "== $t"; cut -d, -f6- gpurun_out/pmc_$t/summary.csv; grep "spmm_" gpurun_out/pmc_$t/kernel_stats.csv | cut -c1-200; done
mkdir -p gpurun_out/r03
python bench.py > gpurun_out/r03/bench_round3.log 2> gpurun_out/r03/bench_round3.err
tail -c 3000 gpurun_out/r03/bench_round3.log
