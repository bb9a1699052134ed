#!/bin/bash
# Memory-side counters of the bench workload: how many L2 misses the Infinity Cache absorbs (RDREQ vs RDREQ_DRAM),
# the average outstanding-read level (latency = LEVEL / RDREQ), write side. Separate --pmc passes.
set -x
mkdir -p gpurun_out/prof_fabric
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
BENCH="python bench.py --steps 50 --warmup 5 --no-extra --no-cpu-baseline"
P=/tmp/pf; O=gpurun_out/prof_fabric
rm -rf $P; mkdir -p $P
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_BUSY_sum TCC_CYCLE_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_TAG_STALL_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_EA0_RDREQ_IO_CREDIT_STALL_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $P/$i -o c -- $BENCH > $O/pass$i.log 2>&1
  f=$(find $P/$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then head -1 $f > $O/pass$i.csv; grep -i "spmm_" $f >> $O/pass$i.csv; fi
done
python scripts/summarize_pmc.py $O/summary.csv $O/pass*.csv
cut -d, -f6- $O/summary.csv
rm -f $O/pass*.csv
