export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
  rm -rf /tmp/pf; timeout 600 rocprofv3 --pmc $set --output-format csv -d /tmp/pf -o c -- python scripts/plan_bench.py --graphs products-sbm --only-plan --iters 3 > /tmp/pf.log 2>&1
  python - $(find /tmp/pf -name "*counter_collection.csv") <<'PY'
import csv, sys, collections
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if "stream_kernel" in r["Kernel_Name"]:
        acc.setdefault((r["Kernel_Name"][13:70], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, "%.0f" % (sum(v) / len(v)), len(v))
PY
done
grep "clustered plan" /tmp/pf.log | cut -c1-120
