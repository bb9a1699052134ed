#!/bin/bash
# Round 5, call ae: 64 KB against 80 KB blocks again, interleaved three times (two separate runs disagreed on the small-world graph by more
# than the effect: box state between processes).
set -x
export TMPDIR=/tmp
O=gpurun_out/r05ae; mkdir -p $O
for round in 1 2 3; do for kb in 4 5; do
  GESPMM_STAGED_LDS_KB=$kb timeout 1500 python scripts/kernel_ab.py --graphs com-amazon-sbm geometric nws-k10 lfr-mu0.1 products-sbm --widths 128 256 512 --kernels staged --tag "round=$round lds_kb=$kb " 2>&1 | grep -v amdgpu >> $O/staged_lds5_interleaved.log
done; done
python - <<'PY'
import re,collections
best=collections.defaultdict(lambda:[1e9,1e9]); allv=collections.defaultdict(list)
for l in open('gpurun_out/r05ae/staged_lds5_interleaved.log'):
    m=re.match(r'round=(\d) lds_kb=(\d) (\S+)\s+N=(\d+).*staged ([\d.]+) us',l)
    if m:
        kb=int(m.group(2)); k=(m.group(3),int(m.group(4))); v=float(m.group(5))
        best[k][kb-4]=min(best[k][kb-4],v); allv[(k,kb)].append(v)
for k in sorted(best): print("%-16s N=%-4d  64 KB min %8.1f (%s)   80 KB min %8.1f (%s)   x%.3f"%(k[0],k[1],best[k][0],"/".join("%.0f"%x for x in allv[(k,4)]),best[k][1],"/".join("%.0f"%x for x in allv[(k,5)]),best[k][1]/best[k][0]))
PY
