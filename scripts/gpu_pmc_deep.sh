#!/bin/bash
# (TA_* counters are left out: that group aborts rocprofv3 on this image.)
# Deeper counter passes (latencies, stalls, address translation) over one command, one rocprofv3 --pmc run per group;
# prints "counter mean-per-launch" for kernels whose name matches the regex.
#   scripts/gpu_pmc_deep.sh <tag> <kernel-name-regex> -- <command...>
TAG=$1; KRE=$2; shift 3
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmcdeep_$TAG.txt; : > $O
i=0
for set in \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
  "TCP_TCP_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
  "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum" \
  "TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_LATENCY_FIFO_FULL_sum TCC_CYCLE_sum" \
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  "TCC_READ_sum TCC_WRITE_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); P=/tmp/pmcdeep_${TAG}_$i; rm -rf $P
  timeout 240 rocprofv3 --pmc $set --output-format csv -d $P -o c -- "$@" > /tmp/pmcdeep_$TAG.log 2>&1
  f=$(find $P -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "  (no output for: $set)" >> $O; tail -3 /tmp/pmcdeep_$TAG.log >> $O; continue; }
  python - "$f" "$KRE" >> $O <<'PY'
import csv, sys, collections, re
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Kernel_Name"]):
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("  %-44s %16.1f  (%d launches)" % (k, sum(v) / len(v), len(v)))
PY
done
echo "== $TAG: $*"; cat $O
