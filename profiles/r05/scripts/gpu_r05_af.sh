#!/bin/bash
# Round 5, call af: the lane-group staged kernel (N = 32 / 64) with 80 KB blocks, interleaved with 64 KB; rows per block 512 / 640.
set -x
export TMPDIR=/tmp
O=gpurun_out/r05af; mkdir -p $O
for round in 1 2 3; do for cfg in "4 0" "5 0" "5 640"; do
  set -- $cfg
  if [ $2 = 0 ]; then unset GESPMM_STAGED_NARROW_ROWS; else export GESPMM_STAGED_NARROW_ROWS=$2; fi
  GESPMM_STAGED_LDS_KB=$1 timeout 1500 python scripts/kernel_ab.py --graphs geometric nws-k10 products-sbm com-amazon-sbm --widths 32 --kernels staged --tag "round=$round lds_kb=$1 rows=$2 " 2>&1 | grep -v amdgpu >> $O/narrow_lds5_interleaved.log
done; done
unset GESPMM_STAGED_NARROW_ROWS
for round in 1 2 3; do for kb in 4 5; do
  GESPMM_STAGED_LDS_KB=$kb timeout 1500 python scripts/kernel_ab.py --graphs geometric nws-k10 products-sbm --widths 64 --kernels staged --tag "round=$round lds_kb=$kb rows=0 " 2>&1 | grep -v amdgpu >> $O/narrow_lds5_interleaved.log
done; done
python - <<'PY'
import re,collections
allv=collections.defaultdict(list)
for l in open('gpurun_out/r05af/narrow_lds5_interleaved.log'):
    m=re.match(r'round=(\d) lds_kb=(\d) rows=(\d+) (\S+)\s+N=(\d+).*staged ([\d.]+) us.*share=([\d.]+)',l)
    if m: allv[(m.group(4),int(m.group(5)),int(m.group(2)),int(m.group(3)))].append((float(m.group(6)),m.group(7)))
for k in sorted(allv): print("%-16s N=%-3d kb=%d rows=%-4d min %8.1f (%s) share %s"%(k[0],k[1],k[2],k[3],min(v for v,_ in allv[k]),"/".join("%.0f"%v for v,_ in allv[k]),allv[k][0][1]))
PY
