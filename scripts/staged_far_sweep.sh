#!/bin/bash
# staged-rows kernel: which columns count as "far" (gathered with nt)? GESPMM_STAGED_FAR_BLOCKS = distance in 128-row blocks, 0 = none
cd $GRAFT_REPO_ROOT
for d in 0 16 64 128 512 2048; do echo "== GESPMM_STAGED_FAR_BLOCKS=$d"; GESPMM_STAGED_FAR_BLOCKS=$d timeout 300 python scripts/staged_time.py products-sbm com-amazon-sbm --n=128,256 2>&1 | grep "kernel=auto\|kernel=staged " | cut -c1-150; done
