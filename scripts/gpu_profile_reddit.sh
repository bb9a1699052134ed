#!/bin/bash
# rocprofv3 on the reddit-like N=128 configuration (slab-blocked path): kernel stats + PMC passes.
set -x
mkdir -p gpurun_out/prof_reddit
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
CMD="python profiles/r01/scripts/reddit_auto_path_for_rocprof.py reddit-like 128 3"
P=/tmp/prof
O=gpurun_out/prof_reddit
rm -rf $P; mkdir -p $P
rocprofv3 -L 2>/dev/null | grep -o -E "\b(TCP|TA|TCC|TD|SQ)_[A-Za-z0-9_]+" | sort -u > $O/counters_available.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -o stats -- $CMD > $O/stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $P/fetch -o fetch -- $CMD > $O/fetch.log 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $P/tcc -o tcc -- $CMD > $O/tcc.log 2>&1
timeout 600 rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --output-format csv -d $P/ta -o ta -- $CMD > $O/ta.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $P/sq -o sq -- $CMD > $O/sq.log 2>&1
for f in $(find $P/stats -name "*stats*.csv"); do cp $f $O/; done
for d in fetch tcc ta sq; do
  for f in $(find $P/$d -name "*counter_collection.csv"); do
    head -1 $f > $O/${d}_counters.csv
    grep -i "spmm_" $f | head -400 >> $O/${d}_counters.csv
  done
done
for f in $(find $P/stats -name "*kernel_trace.csv"); do head -1 $f > $O/kernel_trace_spmm.csv; grep -i "spmm_" $f >> $O/kernel_trace_spmm.csv; done
du -sh gpurun_out
tail -2 $O/*.log
