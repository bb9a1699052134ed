#!/bin/bash
# Round-6 evidence on the final sources: (1) PMC captures behind every traffic figure bench.py quotes (headline graph at N = 128 / 32 / 512
# through its plan, plain call, the structureless stand-in through a steady-state plan, products-shaped communities at N = 128 and 512),
# stamped into profiles/hbm_traffic.json; fresh captures of the reddit-shaped community graph (round-5 review, item 6: the only reddit
# counters were round 1's) and of the RMAT shard (item 1); the general staged kernel at N = 100 / 200; (2) kernel-trace stats of the default
# bench run; (3) the bench line; (4) the several-GPU mode as a one-rank RCCL run; (5) plan times; (6) the GPU suite.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06f; mkdir -p $O
B="python bench.py --no-extra --no-cpu-baseline --steps 50 --warmup 5"
K="spmm_(seg)?stream|spmm_staged|spmm_records|spmm_longrow|spmm_slab"
scripts/gpu_pmc.sh bench_sbm_plan "$K" -- $B > $O/pmc_1.log 2>&1
scripts/gpu_pmc.sh bench_sbm_plain "$K" -- $B --no-plan > $O/pmc_2.log 2>&1
scripts/gpu_pmc.sh bench_like_plan "$K" -- $B --graph com-amazon-like --expected-launches 1000000 > $O/pmc_3.log 2>&1
scripts/gpu_pmc.sh bench_like_plain "$K" -- $B --graph com-amazon-like --no-plan > $O/pmc_4.log 2>&1
scripts/gpu_pmc.sh bench_sbm_plan_N32 "$K" -- $B --ncols 32 --expected-launches 1000000 > $O/pmc_5.log 2>&1
scripts/gpu_pmc.sh bench_sbm_plan_N512 "$K" -- $B --ncols 512 > $O/pmc_6.log 2>&1
scripts/gpu_pmc.sh bench_sbm_plan_N64 "$K" -- $B --ncols 64 --expected-launches 1000000 > $O/pmc_6a.log 2>&1
scripts/gpu_pmc.sh bench_sbm_plan_N100 "$K" -- $B --ncols 100 --expected-launches 1000000 > $O/pmc_6b.log 2>&1
scripts/gpu_pmc.sh bench_sbm_plan_N200 "$K" -- $B --ncols 200 --expected-launches 1000000 > $O/pmc_6c.log 2>&1
scripts/gpu_pmc.sh products_sbm_staged "spmm_staged" -- python scripts/kernel_pmc_case.py products-sbm 128 auto 3 > $O/pmc_7.log 2>&1
scripts/gpu_pmc.sh products_sbm_staged_N512 "spmm_staged" -- python scripts/kernel_pmc_case.py products-sbm 512 auto 3 > $O/pmc_8.log 2>&1
scripts/gpu_pmc.sh reddit_sbm_plan "$K" -- python scripts/kernel_pmc_case.py reddit-sbm 128 auto 3 > $O/pmc_9.log 2>&1
scripts/gpu_pmc.sh rmat24_plain_N256 "$K" -- python scripts/kernel_pmc_case.py rmat-24 256 plain 3 > $O/pmc_10.log 2>&1
python scripts/update_traffic_json.py \
  com-amazon-sbm/N128/valued/plan=gpurun_out/pmc_bench_sbm_plan/summary.csv com-amazon-sbm/N128/valued/plain=gpurun_out/pmc_bench_sbm_plain/summary.csv \
  com-amazon-like/N128/valued/plan=gpurun_out/pmc_bench_like_plan/summary.csv com-amazon-like/N128/valued/plain=gpurun_out/pmc_bench_like_plain/summary.csv \
  com-amazon-sbm/N32/valued/plan=gpurun_out/pmc_bench_sbm_plan_N32/summary.csv com-amazon-sbm/N512/valued/plan=gpurun_out/pmc_bench_sbm_plan_N512/summary.csv \
  com-amazon-sbm/N100/valued/plan=gpurun_out/pmc_bench_sbm_plan_N100/summary.csv com-amazon-sbm/N200/valued/plan=gpurun_out/pmc_bench_sbm_plan_N200/summary.csv \
  products-sbm/N128/valued/plan=gpurun_out/pmc_products_sbm_staged/summary.csv products-sbm/N512/valued/plan=gpurun_out/pmc_products_sbm_staged_N512/summary.csv \
  > $O/update_traffic.log 2>&1
sed -i "s#gpurun_out/pmc_#profiles/r06/pmc_#g" profiles/hbm_traffic.json; cp profiles/hbm_traffic.json $O/hbm_traffic.json
for t in bench_sbm_plan bench_sbm_plain bench_like_plan bench_like_plain bench_sbm_plan_N32 bench_sbm_plan_N64 bench_sbm_plan_N512 bench_sbm_plan_N100 bench_sbm_plan_N200 products_sbm_staged products_sbm_staged_N512 reddit_sbm_plan rmat24_plain_N256; do
  echo "== $t"; grep -E "FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum|TCC_MISS_sum|TCC_EA0_RDREQ_sum" gpurun_out/pmc_$t/summary.csv | cut -d, -f1,6- | cut -c1-200; grep -E "spmm_" gpurun_out/pmc_$t/kernel_stats.csv | cut -c1-200; done > $O/pmc_digest.log 2>&1
P=/tmp/prof_bench; rm -rf $P; mkdir -p $P
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o b -- python bench.py > $O/bench_under_profiler.log 2>&1
f=$(find $P -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_default_kernel_stats.csv
( time python bench.py > $O/bench_round6.log 2> $O/bench_round6.err ) 2> $O/bench_round6.time
cp profiles/bench_extra_last.json $O/bench_extra_round6.json
( time python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --graph rmat --rmat-scale 24 --steps 5 --warmup 2 ) > $O/bench_rmat24_torchrun1.log 2>&1
timeout 600 python scripts/plan_ms.py com-amazon-sbm com-amazon-like pubmed-like products-sbm 2>&1 | grep -v amdgpu | cut -c1-400 > $O/plan_ms.log
timeout 2400 python -m pytest tests -m gpu -q -rs > $O/pytest_gpu_final.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_final.log
timeout 900 python scripts/records_soak.py 9000 300 2>&1 | grep -v amdgpu > $O/records_soak.log
grep "^{" $O/bench_round6.log | cut -c1-4000; cat $O/bench_round6.time; grep "^{" $O/bench_rmat24_torchrun1.log | cut -c1-1500
cat $O/pmc_digest.log | head -120
tail -4 $O/pytest_gpu_final.log | cut -c1-300
