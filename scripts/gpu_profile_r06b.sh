cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06f; mkdir -p $O
K="spmm_(seg)?stream|spmm_staged|spmm_records|spmm_longrow|spmm_slab"
scripts/gpu_pmc.sh products_sbm_staged "spmm_staged" -- python scripts/kernel_pmc_case.py products-sbm 128 auto 3 > $O/pmc_7.log 2>&1
scripts/gpu_pmc.sh products_sbm_staged_N512 "spmm_staged" -- python scripts/kernel_pmc_case.py products-sbm 512 auto 3 > $O/pmc_8.log 2>&1
scripts/gpu_pmc.sh reddit_sbm_plan "$K" -- python scripts/kernel_pmc_case.py reddit-sbm 128 auto 3 > $O/pmc_9.log 2>&1
scripts/gpu_pmc.sh rmat24_plain_N256 "$K" -- python scripts/kernel_pmc_case.py rmat-24 256 plain 3 > $O/pmc_10.log 2>&1
for t in products_sbm_staged products_sbm_staged_N512 reddit_sbm_plan rmat24_plain_N256; do echo "== $t"; grep -E "FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum|TCC_MISS_sum" gpurun_out/pmc_$t/summary.csv | grep -v ",1,[0-9.]*$" | cut -d, -f1,2,6- | cut -c1-220; done
