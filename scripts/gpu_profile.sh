#!/bin/bash
# rocprofv3: kernel-trace stats, then PMC passes (separate runs, as the guide prescribes).
# Raw output goes to /tmp on the GPU box; only small CSV summaries are copied to gpurun_out/.
set -x
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
BENCH="python bench.py --steps 50 --warmup 5 --no-extra --no-cpu-baseline"
P=/tmp/prof
rm -rf $P; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -o stats -- $BENCH > gpurun_out/prof/stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch -o fetch -- $BENCH > gpurun_out/prof/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write -o write -- $BENCH > gpurun_out/prof/write.log 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $P/tcc -o tcc -- $BENCH > gpurun_out/prof/tcc.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $P/sq -o sq -- $BENCH > gpurun_out/prof/sq.log 2>&1
find $P -type f | xargs ls -la | head -60
for f in $(find $P -name "*stats*.csv"); do cp $f gpurun_out/prof/; done
# counter CSVs: keep only the rows of our kernels
for d in fetch write tcc sq; do
  for f in $(find $P/$d -name "*counter_collection.csv"); do
    head -1 $f > gpurun_out/prof/${d}_counters_spmm.csv
    grep -i "spmm_" $f >> gpurun_out/prof/${d}_counters_spmm.csv
  done
done
for f in $(find $P/stats -name "*kernel_trace.csv"); do head -1 $f > gpurun_out/prof/kernel_trace_spmm.csv; grep -i "spmm_" $f >> gpurun_out/prof/kernel_trace_spmm.csv; done
timeout 900 python scripts/ksweep.py > gpurun_out/ksweep.log 2>&1
du -sh gpurun_out
