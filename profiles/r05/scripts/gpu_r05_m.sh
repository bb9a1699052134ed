#!/bin/bash
# Round 5, call m: one PMC pass over RMAT-24 x 256 (the per-GPU shard shape of config 5) — L2 hits, fabric bytes — plain call and default plan.
set -x
export TMPDIR=/tmp
bash scripts/gpu_pmc.sh rmat24_plain_N256 "spmm" -- python scripts/kernel_pmc_case.py rmat-24 256 plain 3
bash scripts/gpu_pmc.sh rmat24_auto_N256 "spmm" -- python scripts/kernel_pmc_case.py rmat-24 256 auto 3
