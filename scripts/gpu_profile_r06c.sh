#!/bin/bash
# Round-6 closing evidence on the FINAL sources (the staged kernel gained its continuing form and the plans their column-slab tables after
# scripts/gpu_profile_r06.sh ran: bench.py quotes only traffic captured at the tree's own fingerprint). The same captures as
# gpu_profile_r06.sh without the RMAT shard and the N = 64 width (not quoted by bench.py); reddit_sbm_plan is now the slab launches.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06c; mkdir -p $O
B="python bench.py --no-extra --no-cpu-baseline --steps 50 --warmup 5"
K="spmm_(seg)?stream|spmm_staged|spmm_records|spmm_longrow|spmm_slab"
scripts/gpu_pmc.sh bench_sbm_plan "$K" -- $B > $O/pmc_1.log 2>&1
scripts/gpu_pmc.sh bench_sbm_plain "$K" -- $B --no-plan > $O/pmc_2.log 2>&1
scripts/gpu_pmc.sh bench_like_plan "$K" -- $B --graph com-amazon-like --expected-launches 1000000 > $O/pmc_3.log 2>&1
scripts/gpu_pmc.sh bench_like_plain "$K" -- $B --graph com-amazon-like --no-plan > $O/pmc_4.log 2>&1
scripts/gpu_pmc.sh bench_sbm_plan_N32 "$K" -- $B --ncols 32 --expected-launches 1000000 > $O/pmc_5.log 2>&1
scripts/gpu_pmc.sh bench_sbm_plan_N512 "$K" -- $B --ncols 512 > $O/pmc_6.log 2>&1
scripts/gpu_pmc.sh bench_sbm_plan_N100 "$K" -- $B --ncols 100 --expected-launches 1000000 > $O/pmc_6b.log 2>&1
scripts/gpu_pmc.sh bench_sbm_plan_N200 "$K" -- $B --ncols 200 --expected-launches 1000000 > $O/pmc_6c.log 2>&1
scripts/gpu_pmc.sh products_sbm_staged "spmm_staged" -- python scripts/kernel_pmc_case.py products-sbm 128 auto 3 > $O/pmc_7.log 2>&1
scripts/gpu_pmc.sh products_sbm_staged_N512 "spmm_staged" -- python scripts/kernel_pmc_case.py products-sbm 512 auto 3 > $O/pmc_8.log 2>&1
scripts/gpu_pmc.sh reddit_sbm_plan "$K" -- python scripts/kernel_pmc_case.py reddit-sbm 128 auto 3 > $O/pmc_9.log 2>&1
python scripts/update_traffic_json.py \
  com-amazon-sbm/N128/valued/plan=gpurun_out/pmc_bench_sbm_plan/summary.csv com-amazon-sbm/N128/valued/plain=gpurun_out/pmc_bench_sbm_plain/summary.csv \
  com-amazon-like/N128/valued/plan=gpurun_out/pmc_bench_like_plan/summary.csv com-amazon-like/N128/valued/plain=gpurun_out/pmc_bench_like_plain/summary.csv \
  com-amazon-sbm/N32/valued/plan=gpurun_out/pmc_bench_sbm_plan_N32/summary.csv com-amazon-sbm/N512/valued/plan=gpurun_out/pmc_bench_sbm_plan_N512/summary.csv \
  com-amazon-sbm/N100/valued/plan=gpurun_out/pmc_bench_sbm_plan_N100/summary.csv com-amazon-sbm/N200/valued/plan=gpurun_out/pmc_bench_sbm_plan_N200/summary.csv \
  products-sbm/N128/valued/plan=gpurun_out/pmc_products_sbm_staged/summary.csv products-sbm/N512/valued/plan=gpurun_out/pmc_products_sbm_staged_N512/summary.csv \
  > $O/update_traffic.log 2>&1
sed -i "s#gpurun_out/pmc_#profiles/r06/pmc_#g" profiles/hbm_traffic.json; cp profiles/hbm_traffic.json $O/hbm_traffic.json
P=/tmp/prof_bench; rm -rf $P; mkdir -p $P
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o b -- python bench.py > $O/bench_under_profiler.log 2>&1
f=$(find $P -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_default_kernel_stats.csv
( time python bench.py > $O/bench_round6.log 2> $O/bench_round6.err ) 2> $O/bench_round6.time
cp profiles/bench_extra_last.json $O/bench_extra_round6.json
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 2400 python -m pytest tests -m gpu -q -rs > $O/pytest_gpu_final.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_final.log
grep "^{" $O/bench_round6.log | cut -c1-3000; cat $O/bench_round6.time
tail -n 4 $O/pytest_gpu_final.log | cut -c1-300; tail -n 1 $O/smoke.log | cut -c1-300
