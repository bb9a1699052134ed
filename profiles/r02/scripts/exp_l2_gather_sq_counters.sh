export TMPDIR=/tmp
for f in 0x180 0x120; do
  rm -rf /tmp/pm_$f
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm_$f -o c -- python profiles/r02/scripts/fetch_calibration.py --m 232965 --degs 64 --k 4096 --flags $f --iters 20 > /dev/null 2>&1
  rm -rf /tmp/pn_$f
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d /tmp/pn_$f -o c -- python profiles/r02/scripts/fetch_calibration.py --m 232965 --degs 64 --k 4096 --flags $f --iters 20 > /dev/null 2>&1
  echo "== flags $f"
  for d in /tmp/pm_$f /tmp/pn_$f; do python - $(find $d -name "*counter_collection.csv") <<'PY'
import csv, sys, collections
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if "stream_kernel" in r["Kernel_Name"]:
        acc.setdefault((r["Kernel_Name"][13:60], r["Counter_Name"], r["VGPR_Count"], r["LDS_Block_Size"]), []).append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("  ", k, "%.0f" % (sum(v) / len(v)))
PY
  done
done
