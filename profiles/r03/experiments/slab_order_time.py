"""Cache-blocked path on the reddit-shaped graph: rows in storage order vs longest-first (plan flag 0x40000 = storage order).
python scripts/slab_order_time.py [N ...]"""
import statistics, sys
import torch
sys.path.insert(0, ".")
import gespmm_amd
from gespmm_amd import graphs, spmm

g = graphs.synthetic_graph("reddit-like", seed=42, device="cuda")
val = torch.rand(g["nnz"], device="cuda") - 0.5
for N in [int(x) for x in sys.argv[1:]] or [64, 128, 256]:
    B = torch.rand(g["K"], N, device="cuda") - 0.5
    ref = spmm.csr_spmm(g["rowptr"], g["colind"], val, B)
    for label, flags in (("storage order", 0x40000), ("longest first", 0)):
        plan = spmm.SpmmPlan(g["rowptr"], g["colind"], g["K"], N, values=val, flags=flags)
        C = torch.empty_like(ref)
        for _ in range(3): spmm.csr_spmm(g["rowptr"], g["colind"], val, B, out=C, plan=plan)
        same = bool(torch.equal(C.view(torch.int32), ref.view(torch.int32)))
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, b in ev:
            a.record(); spmm.csr_spmm(g["rowptr"], g["colind"], val, B, out=C, plan=plan); b.record()
        torch.cuda.synchronize()
        print("N=%d %-14s %8.1f us  bits equal plain call: %s | %s" % (N, label, statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev), same, plan.describe()[:110]), flush=True)
