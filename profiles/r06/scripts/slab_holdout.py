"""Column-slab tables on the dense LFR hold-out graphs (networkx, mean degree ~ 325; scripts/holdout_graphs.py): AUTO, AUTO without slab
tables (second process, GESPMM_SLABS=-1) and the tables asked for by name on the clustered order.
    GESPMM_HOLDOUT_DIR=/tmp/holdout python scripts/holdout_graphs.py lfr-verydense-mu0.2 && python profiles/r06/scripts/slab_holdout.py lfr-verydense-mu0.2"""
import os, statistics, sys
import numpy as np
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "scripts")
import gespmm_amd
from gespmm_amd import _lib, spmm
import holdout_audit

name = sys.argv[1]
g = holdout_audit.from_npz(os.path.join(os.environ.get("GESPMM_HOLDOUT_DIR", "/tmp/holdout"), name + ".npz"))
rp, ci, M, K, nnz = g["rowptr"], g["colind"], g["M"], g["K"], g["nnz"]
val = torch.rand(nnz, device="cuda") - 0.5
B = torch.rand(K, 128, device="cuda") - 0.5
C = torch.empty(M, 128, device="cuda")
ref = spmm.csr_spmm(rp, ci, val, B, cfg={"flags": _lib.FLAG_STRICT_ORDER})
for kern, reorder in (("auto", "auto"), ("seg-stream", True), ("staged-slabs", True)):
    plan = spmm.SpmmPlan(rp, ci, K, 128, values=val, kernel=kern, reorder=reorder)
    fn = lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan)
    for _ in range(2): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(9)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev)
    d = plan.describe()
    print("%s M=%d mean degree %.0f GESPMM_SLABS=%s kernel=%-12s reorder=%-5s %8.1f us bits=%s | %s | %s" % (
        name, M, nnz / M, os.environ.get("GESPMM_SLABS", "-"), kern, reorder, t, bool(torch.equal(C.view(torch.int32), ref.view(torch.int32))),
        d.split("|")[0].strip()[:110], d.split("|")[-1].strip()[:110]), flush=True)
    del plan
