#!/bin/bash
# Where the 3.6 ms of a plan go (headline graph): the plan's own lap timer (synchronising) and a kernel trace of 8 creations.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/plan_profile; mkdir -p $O
cd $R
GESPMM_PLAN_TIMING=1 python scripts/plan_ms.py --reps 3 com-amazon-sbm > $O/plan_laps.log 2>&1
python scripts/plan_ms.py --reps 7 com-amazon-sbm > $O/plan_ms.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o plan -- python scripts/plan_ms.py --reps 7 com-amazon-sbm > $O/trace.log 2>&1
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/trace -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/trace
tail -3 $O/plan_ms.log
