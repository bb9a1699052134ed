"""The 41-class layer of the config-4 GCN on the dense community graph: the plan's product at N = 41 against N = 44 / 48 / 64 (the width a caller
could pad its features to; the first 41 columns have the same bits).   python profiles/r06/scripts/dense_odd_width.py [graph]"""
import statistics, sys
import torch
sys.path.insert(0, ".")
import gespmm_amd
from gespmm_amd import graphs, spmm

name = sys.argv[1] if len(sys.argv) > 1 else "reddit-sbm"
g = graphs.synthetic_graph(name, seed=42, device="cuda")
rp, ci, M, K, nnz = g["rowptr"], g["colind"], g["M"], g["K"], g["nnz"]
val = torch.rand(nnz, device="cuda") - 0.5
B64 = torch.rand(K, 64, device="cuda") - 0.5
ref = None
for N, variant in ((41, -1), (44, -1), (44, 3), (48, -1), (48, 3), (64, -1), (64, 3)):
    B = torch.zeros(K, N, device="cuda")
    B[:, :41] = B64[:, :41]
    C = torch.empty(M, N, device="cuda")
    plan = spmm.SpmmPlan(rp, ci, K, N, values=val, variant=variant)
    fn = lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan, variant=variant)
    for _ in range(2): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev)
    if ref is None: ref = C[:, :41].clone()
    same = bool(torch.equal(C[:, :41].contiguous().view(torch.int32), ref.contiguous().view(torch.int32)))
    print("%s N=%d variant=%d %8.1f us first 41 columns bits=%s | %s" % (name, N, variant, t, same, plan.describe().split("|")[-1].strip()[:130]), flush=True)
    del plan
