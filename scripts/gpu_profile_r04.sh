#!/bin/bash
# Round-4 evidence on the final sources: (1) PMC captures behind every traffic figure bench.py quotes (headline graph at N = 128 /
# 32 / 512 through its plan, plain call, the structureless stand-in, products-shaped communities at N = 128 and 512), stamped into
# profiles/hbm_traffic.json; (2) kernel-trace stats of the default bench run; (3) the bench line; (4) plan audit against round 3's
# log; (5) hold-out and stand-in audits; (6) the GPU suite.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
B="python bench.py --no-extra --no-cpu-baseline --steps 50 --warmup 5"
K="spmm_(seg)?stream|spmm_staged"
scripts/gpu_pmc.sh bench_sbm_plan "$K" -- $B > gpurun_out/r04/pmc_1.log 2>&1
scripts/gpu_pmc.sh bench_sbm_plain "$K" -- $B --no-plan > gpurun_out/r04/pmc_2.log 2>&1
scripts/gpu_pmc.sh bench_like_plan "$K" -- $B --graph com-amazon-like > gpurun_out/r04/pmc_3.log 2>&1
scripts/gpu_pmc.sh bench_like_plain "$K" -- $B --graph com-amazon-like --no-plan > gpurun_out/r04/pmc_4.log 2>&1
scripts/gpu_pmc.sh bench_sbm_plan_N32 "$K" -- $B --ncols 32 > gpurun_out/r04/pmc_5.log 2>&1
scripts/gpu_pmc.sh bench_sbm_plan_N512 "$K" -- $B --ncols 512 > gpurun_out/r04/pmc_6.log 2>&1
scripts/gpu_pmc.sh products_sbm_staged "spmm_staged" -- python profiles/r03/experiments/narrow_rows_sbm.py 128 > gpurun_out/r04/pmc_7.log 2>&1
scripts/gpu_pmc.sh products_sbm_staged_N512 "spmm_staged" -- python profiles/r03/experiments/narrow_rows_sbm.py 512 > gpurun_out/r04/pmc_8.log 2>&1
python scripts/update_traffic_json.py \
  com-amazon-sbm/N128/valued/plan=gpurun_out/pmc_bench_sbm_plan/summary.csv com-amazon-sbm/N128/valued/plain=gpurun_out/pmc_bench_sbm_plain/summary.csv \
  com-amazon-like/N128/valued/plan=gpurun_out/pmc_bench_like_plan/summary.csv com-amazon-like/N128/valued/plain=gpurun_out/pmc_bench_like_plain/summary.csv \
  com-amazon-sbm/N32/valued/plan=gpurun_out/pmc_bench_sbm_plan_N32/summary.csv com-amazon-sbm/N512/valued/plan=gpurun_out/pmc_bench_sbm_plan_N512/summary.csv \
  products-sbm/N128/valued/plan=gpurun_out/pmc_products_sbm_staged/summary.csv products-sbm/N512/valued/plan=gpurun_out/pmc_products_sbm_staged_N512/summary.csv \
  > gpurun_out/r04/update_traffic.log 2>&1
sed -i "s#gpurun_out/pmc_#profiles/r04/pmc_#g" profiles/hbm_traffic.json; cp profiles/hbm_traffic.json gpurun_out/r04/hbm_traffic.json
for t in bench_sbm_plan bench_sbm_plain bench_like_plan bench_like_plain bench_sbm_plan_N32 bench_sbm_plan_N512 products_sbm_staged products_sbm_staged_N512; do
  echo "== $t"; grep -E "FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum|TCC_MISS_sum|TCC_EA0_RDREQ_sum" gpurun_out/pmc_$t/summary.csv | cut -d, -f1,6- | cut -c1-200; grep -E "spmm_" gpurun_out/pmc_$t/kernel_stats.csv | cut -c1-200; done
P=/tmp/prof_bench; rm -rf $P; mkdir -p $P
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o b -- python bench.py > gpurun_out/r04/bench_under_profiler.log 2>&1
f=$(find $P -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r04/bench_default_kernel_stats.csv
( time python bench.py > gpurun_out/r04/bench_round4.log 2> gpurun_out/r04/bench_round4.err ) 2> gpurun_out/r04/bench_round4.time
cp profiles/bench_extra_last.json gpurun_out/r04/bench_extra_round4.json
# the multi-GPU mode as a one-rank RCCL run (the form the driver launches for N > 1), RMAT scale 24
( time python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --graph rmat --rmat-scale 24 --steps 5 --warmup 2 ) > gpurun_out/r04/bench_rmat24_torchrun1.log 2>&1
timeout 1500 python scripts/plan_audit.py --baseline profiles/r03/plan_audit.log > gpurun_out/r04/plan_audit.log 2>&1; echo "plan_audit rc=$?" >> gpurun_out/r04/plan_audit.log
if [ -d profiles/r04/holdout ] && ls profiles/r04/holdout/*.npz > /dev/null 2>&1; then
  timeout 1800 python scripts/holdout_audit.py 2>&1 | grep -v "amdgpu.ids\|^W2026" > gpurun_out/r04/holdout_audit.log
fi
timeout 1800 python scripts/holdout_audit.py --standins --widths 32 64 128 256 512 2>&1 | grep -v "amdgpu.ids\|^W2026" > gpurun_out/r04/standin_audit.log
timeout 2400 python -m pytest tests -m gpu -q -rs > gpurun_out/r04/pytest_gpu_final.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/pytest_gpu_final.log
tail -1 gpurun_out/r04/bench_round4.log | cut -c1-4000; cat gpurun_out/r04/bench_round4.time; grep "^{" gpurun_out/r04/bench_rmat24_torchrun1.log | cut -c1-1500
tail -4 gpurun_out/r04/plan_audit.log | cut -c1-300; grep "<--\|worst" gpurun_out/r04/holdout_audit.log gpurun_out/r04/standin_audit.log | cut -c1-300; tail -4 gpurun_out/r04/pytest_gpu_final.log | cut -c1-300
