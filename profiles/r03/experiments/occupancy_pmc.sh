#!/bin/bash
# Fabric reads of the scalar-walk experiment kernel (nothing staged) against the number of resident wavefronts per CU.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
P=/tmp/occ_pmc; rm -rf $P; mkdir -p $P
PMC_ONLY=1 timeout 900 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $P -o c -- python profiles/r03/experiments/occupancy_sweep.py products-sbm 256 > $P/out.log 2>&1
grep "task_entries" $P/out.log | cut -c1-140
f=$(find $P -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "scalar_kernel" in r["Kernel_Name"]]
# launch order = the sweep's order: 10 launches per occupancy setting (1 check + 2 warm-up + 7 timed)
by = collections.OrderedDict()
for r in rows:
    by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by)
labels = ["no limit (32 wavefronts/CU)", "<= 8 workgroups (32)", "<= 6 (24)", "<= 4 (16)", "<= 2 (8)"]
for i, lab in enumerate(labels):
    grp = [by[d] for d in ids[10 * i:10 * i + 10]]
    if not grp: continue
    print("%-30s " % lab + "  ".join("%s %.1f M" % (n.replace("_sum", ""), sum(g[n] for g in grp) / len(grp) / 1e6) for n in sorted(grp[0])) + "  (%d launches)" % len(grp))
PY
