#!/bin/bash
# Round 5, call u: the bench line again (its roofline is priced with the region-average launch duration now) with the rocprofv3 kernel
# stats of the same command beside it.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
P=/tmp/benchprof; rm -rf $P
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o b -- python bench.py > $O/bench_under_profiler.log 2>&1
f=$(find $P -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_default_kernel_stats.csv
( time python bench.py > $O/bench_round5.log 2> $O/bench_round5.err ) 2> $O/bench_round5.time
cp profiles/bench_extra_last.json $O/bench_extra_round5.json
grep "^{" $O/bench_round5.log | cut -c1-1500; grep staged_kernel $O/bench_default_kernel_stats.csv | cut -c1-30,100-220
