#!/bin/bash
# Round 5, call al: rows per block at 128 columns once more, now with 160 staged rows per block (80 KB): 80 / 96 / 112 / 128 / 144, two rounds.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05al; mkdir -p $O
for round in 1 2; do for r in 80 96 112 128 144; do
  GESPMM_STAGED_ROWS=$r timeout 900 python scripts/kernel_ab.py --graphs com-amazon-sbm geometric nws-k10 lfr-mu0.1 products-sbm --widths 128 --kernels staged --tag "round=$round rows=$r " 2>&1 | grep -v amdgpu >> $O/staged_rows_128_lds5.log
done; done
python - <<'PY'
import re,collections
allv=collections.defaultdict(list)
for l in open('gpurun_out/r05al/staged_rows_128_lds5.log'):
    m=re.match(r'round=(\d) rows=(\d+) (\S+)\s+N=(\d+).*staged ([\d.]+) us.*share=([\d.]+)',l)
    if m: allv[(m.group(3),int(m.group(2)))].append((float(m.group(5)),m.group(6)))
for k in sorted(allv): print("%-16s rows=%-4d min %8.1f (%s) share %s"%(k[0],k[1],min(v for v,_ in allv[k]),"/".join("%.0f"%v for v,_ in allv[k]),allv[k][0][1]))
PY
