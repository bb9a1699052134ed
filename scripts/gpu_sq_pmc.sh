#!/bin/bash
# Instruction mix / issue-unit activity of one kernel (separate --pmc passes, no tracing beside them).
#   scripts/gpu_sq_pmc.sh <tag> <kernel-name-substring> -- <command...>
TAG=$1; KRE=$2; shift 3
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/sq_$TAG.log; : > $O
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  P=/tmp/sq_pmc_$TAG; rm -rf $P; mkdir -p $P
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $P -o c -- "$@" > $P/out.log 2>&1
  f=$(find $P -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "set [$set]: no output" >> $O; tail -3 $P/out.log >> $O; continue; fi
  python - "$f" "$KRE" >> $O <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if sys.argv[2] in k:
        agg[(k[:44], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print("%-46s %-28s %16.0f per launch (%d launches)" % (k, c, sum(v) / len(v), len(v)))
PY
done
cat $O
