#!/bin/bash
# Round 5, call v: the reference's driver protocol (200 launches per width, rocSPARSE column beside ours) on the stand-ins.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05v; mkdir -p $O
timeout 1200 python scripts/driver_compare.py 2>&1 | grep -v amdgpu > $O/driver_compare.log
cat $O/driver_compare.log
