#!/bin/bash
# FETCH_SIZE / request counters of the calibration launches (one rocprofv3 --pmc pass per counter group).
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/fetch_calibration; mkdir -p $O; : > $O/summary.txt
CASES=${CASES:-"--deg 1|--deg 1 --identity|--deg 2|--deg 3"}
IFS='|' read -ra CL <<< "$CASES"
for c in "${CL[@]}"; do
  tag=$(echo $c | tr -d ' -' | tr ',' '_')
  python profiles/r02/scripts/fetch_calibration.py $c 2>&1 | grep gathers >> $O/summary.txt
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" ${EXTRA_SETS:+"$EXTRA_SETS"}; do
    i=$((i+1)); P=/tmp/cal_${tag}_$i; rm -rf $P
    timeout 600 rocprofv3 --pmc $set --output-format csv -d $P -o c -- python profiles/r02/scripts/fetch_calibration.py $c > /dev/null 2>&1
    f=$(find $P -name "*counter_collection.csv" | head -1)
    python - "$f" "$tag" >> $O/summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "spmm_stream_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("  %-24s %-26s mean/launch %.1f (%d launches)" % (sys.argv[2], k, sum(v) / len(v), len(v)))
PY
  done
done
cat $O/summary.txt
