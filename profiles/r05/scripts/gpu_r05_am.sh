#!/bin/bash
# Round 5, call am: 16 gathers per chunk at N = 128 again, with the retuned blocks (80 KB, rows by mean degree), interleaved.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05am; mkdir -p $O
for round in 1 2 3; do for u in 8 16; do
  GESPMM_STAGED_U=$u timeout 900 python scripts/kernel_ab.py --graphs com-amazon-sbm geometric nws-k10 lfr-mu0.1 products-sbm --widths 128 --kernels staged --tag "round=$round U=$u " 2>&1 | grep -v amdgpu >> $O/staged_u16_lds5.log
done; done
python - <<'PY'
import re,collections
allv=collections.defaultdict(list)
for l in open('gpurun_out/r05am/staged_u16_lds5.log'):
    m=re.match(r'round=(\d) U=(\d+) (\S+)\s+N=(\d+).*staged ([\d.]+) us',l)
    if m: allv[(m.group(3),int(m.group(2)))].append(float(m.group(5)))
for k in sorted(allv): print("%-16s U=%-3d min %8.1f (%s)"%(k[0],k[1],min(allv[k]),"/".join("%.0f"%v for v in allv[k])))
PY
