#!/bin/bash
# Fabric reads / L2 hits of the experiment kernels on products-sbm: scalar walk with nothing staged (mode 4) vs staged (mode 11), same tasks.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
CFG=${CFG:-128,128,16}; MODES=${MODES:-4,11}; export MODES
echo "== block rows, staged rows, wavefronts per block = $CFG; modes $MODES"
for set in "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  P=/tmp/hr_pmc; rm -rf $P; mkdir -p $P
  timeout 900 rocprofv3 --pmc $set --output-format csv -d $P -o c -- python profiles/r03/experiments/hotrows_time.py products-sbm $CFG > /dev/null 2>&1
  f=$(find $P -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "scalar" in k or "segstream" in k:
        agg[(k[:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print("%-72s %-24s %14.0f per launch (%d launches)" % (k, c, sum(v) / len(v), len(v)))
PY
done
