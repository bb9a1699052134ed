#!/bin/bash
# Kernel-time breakdown of the device analysis stage: rocprofv3 --kernel-trace --stats of scripts/plan_time.py.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
G=${1:-com-amazon-sbm}
P=/tmp/prof_plan; rm -rf $P; mkdir -p $P gpurun_out/prof_plan
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o plan -- python scripts/plan_time.py $G > gpurun_out/prof_plan/run_$G.log 2>&1
for f in $(find $P -name "*kernel_stats.csv"); do cp $f gpurun_out/prof_plan/kernel_stats_$G.csv; done
head -40 gpurun_out/prof_plan/kernel_stats_$G.csv
grep -v "^W2026\|^E2026" gpurun_out/prof_plan/run_$G.log | tail -5
