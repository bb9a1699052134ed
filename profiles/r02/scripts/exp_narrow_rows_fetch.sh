export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for n in 16 32 64; do
  rm -rf /tmp/pf; rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o c -- python profiles/r02/scripts/fetch_calibration.py --graph products-like --n $n --iters 3 > /tmp/pf.log 2>&1
  grep gathers /tmp/pf.log | cut -c1-200
  python - $(find /tmp/pf -name "*counter_collection.csv") $n <<'PY'
import csv, sys, collections
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if "spmm_" in r["Kernel_Name"] and "stream_kernel" in r["Kernel_Name"]:
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("  N=%s %s mean/launch %.0f KiB -> x2 = %.2f GB fetched" % (sys.argv[2], k, sum(v)/len(v), 2*sum(v)/len(v)*1024/1e9))
PY
done
