#!/bin/bash
# Instruction mix / issue-unit activity of the staged-rows kernel on products-sbm N = 128 (separate --pmc passes).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES SQ_INSTS_SENDMSG"; do
  P=/tmp/sq_pmc; rm -rf $P; mkdir -p $P
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $P -o c -- python profiles/r03/experiments/narrow_rows_sbm.py 128 ${KERNEL:-auto} > $P/out.log 2>&1
  f=$(find $P -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "set [$set]: no output"; tail -3 $P/out.log; continue; fi
  python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "spmm_staged" in k or "segstream" in k:
        agg[(k[:44], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print("%-46s %-24s %16.0f per launch (%d launches)" % (k, c, sum(v) / len(v), len(v)))
PY
done
