export TMPDIR=/tmp
for f in 0x20 0x80; do
  rm -rf /tmp/st_$f
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$f -o s -- python profiles/r02/scripts/fetch_calibration.py --m 232965 --degs 64 --k 4096 --flags $f --iters 20 > /dev/null 2>&1
  echo "== flags $f"; cut -d, -f1-4 $(find /tmp/st_$f -name "*kernel_stats.csv") | grep -i spmm | head -3
  rm -rf /tmp/pm_$f
  rocprofv3 --pmc SQ_INSTS_VMEM_RD TCP_TOTAL_ACCESSES_sum SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d /tmp/pm_$f -o c -- python profiles/r02/scripts/fetch_calibration.py --m 232965 --degs 64 --k 4096 --flags $f --iters 20 > /dev/null 2>&1
  python - $(find /tmp/pm_$f -name "*counter_collection.csv") <<'PY'
import csv, sys, collections
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if "spmm" in r["Kernel_Name"]:
        acc.setdefault((r["Kernel_Name"][:70], r["Counter_Name"], r["VGPR_Count"], r["Accum_VGPR_Count"] if "Accum_VGPR_Count" in r else "", r["LDS_Block_Size"]), []).append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("  ", k, sum(v) / len(v))
PY
done
