#!/bin/bash
# L1->L2 and L2->fabric request counts of the planned SpMM on products-sbm at N = 16 and N = 32 (separate --pmc passes).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for N in 16 32; do
  for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    P=/tmp/nr_$N; rm -rf $P; mkdir -p $P
    timeout 600 rocprofv3 --pmc $set --output-format csv -d $P -o c -- python profiles/r03/experiments/narrow_rows_sbm.py $N > /dev/null 2>&1
    f=$(find $P -name "*counter_collection.csv" | head -1)
    python - "$f" $N <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "spmm_" in r["Kernel_Name"] and "stream" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in agg.items():
    print("N=%s %-62s %-30s %14.0f per launch (%d launches)" % (sys.argv[2], k, c, sum(v) / len(v), len(v)))
PY
  done
done
