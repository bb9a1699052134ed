#!/bin/bash
# Round 5, call m: one PMC pass over RMAT-24 x 256 (the per-GPU shard shape of config 5) — L2 hits, fabric bytes — plain call and default plan.
set -x
export TMPDIR=/tmp
bash scripts/gpu_pmc.sh rmat24_plain_N256 "spmm" -- python scripts/kernel_pmc_case.py rmat-24 256 plain 3
bash scripts/gpu_pmc.sh rmat24_auto_N256 "spmm" -- python scripts/kernel_pmc_case.py rmat-24 256 auto 3
# which kernels the analysis of the headline graph spends its 5 ms in (three sweeps per level now)
O=gpurun_out/r05m; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/planprof -o p -- python scripts/plan_ms.py com-amazon-sbm --reps 10 > $O/plan_ms_profiled.log 2>&1
f=$(find /tmp/planprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/plan_kernel_stats.csv
GESPMM_PLAN_TIMING=1 timeout 300 python scripts/plan_ms.py com-amazon-sbm geometric --reps 3 > $O/plan_timing.log 2>&1
timeout 600 python scripts/kernel_ab.py --graphs com-amazon-sbm geometric lfr-mu0.1 nws-k10 --widths 128 --kernels staged --tag "sweeps=policy " 2>&1 | grep -v amdgpu >> $O/plan_timing.log
head -40 $O/plan_kernel_stats.csv | cut -c1-200; grep -v amdgpu $O/plan_timing.log | tail -50
