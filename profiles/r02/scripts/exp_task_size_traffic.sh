export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for te in 32 40 48 56; do
  for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    rm -rf /tmp/pf; rocprofv3 --pmc $set --output-format csv -d /tmp/pf -o c -- python scripts/plan_bench.py --graphs com-amazon-sbm --only-plan --iters 50 --task-entries $te > /tmp/pf.log 2>&1
    python - $(find /tmp/pf -name "*counter_collection.csv") $te <<'PY'
import csv, sys, collections
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if "spmm_stream_kernel" in r["Kernel_Name"]:
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
out = {k: sum(v) / len(v) for k, v in acc.items()}
if "FETCH_SIZE" in out: print("entries %s: FETCH %.0f KiB -> %.1f MB fetched, total %.0f MB = %.2fx" % (sys.argv[2], out["FETCH_SIZE"], 2*out["FETCH_SIZE"]*1024/1e6, (2*out["FETCH_SIZE"]*1024 + 171.45e6)/1e6, (2*out["FETCH_SIZE"]*1024 + 171.45e6)/359.05e6))
else: print("entries %s: L2 hit %.3f" % (sys.argv[2], out["TCC_HIT_sum"]/(out["TCC_HIT_sum"]+out["TCC_MISS_sum"])))
PY
  done
  grep "clustered plan" /tmp/pf.log | cut -c1-70
done
