#!/bin/bash
# rocprofv3 passes over one command: kernel-trace stats, then separate --pmc passes (HBM/fabric bytes, L2 hits,
# wave states), condensed to one line per (kernel, counter) by summarize_pmc.py.
#   scripts/gpu_pmc.sh <tag> <kernel-name-regex> -- <command...>
set -x
TAG=$1; KRE=$2; shift 3
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_$TAG; P=/tmp/pmc_$TAG
rm -rf $P; mkdir -p $P $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -o s -- "$@" > $O/stats.log 2>&1
f=$(find $P/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_BUSY_sum TCC_CYCLE_sum"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $set --output-format csv -d $P/$i -o c -- "$@" > $O/pass$i.log 2>&1
  f=$(find $P/$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then head -1 $f > $O/pass$i.csv; grep -E "$KRE" $f >> $O/pass$i.csv; fi
done
python scripts/summarize_pmc.py $O/summary.csv $O/pass*.csv
cat $O/summary.csv | cut -d, -f1-12 | head -60
rm -f $O/pass*.csv
