"""SDDMM: CSR form vs COO form (and through a plan) on short-row graphs, median of 100 launches.
python scripts/sddmm_audit.py [graph ...]"""
import statistics, sys
import torch
sys.path.insert(0, ".")
import gespmm_amd
from gespmm_amd import graphs, sddmm, spmm

def med(fn, n=100):
    for _ in range(5): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev)

for name in sys.argv[1:] or ["com-amazon-like", "com-amazon-sbm", "cit-hepth-like", "pubmed-like"]:
    g = graphs.synthetic_graph(name, seed=42, device="cuda")
    rp, ci, M = g["rowptr"], g["colind"], g["M"]
    rows = torch.repeat_interleave(torch.arange(M, device="cuda", dtype=torch.int32), (rp[1:] - rp[:-1]).long()).to(torch.int32)
    plan = spmm.SpmmPlan(rp, ci, g["K"], 128)
    for N in (16, 32, 41, 64, 128, 256):
        D1 = torch.rand(M, N, device="cuda") - 0.5
        D2 = torch.rand(g["K"], N, device="cuda") - 0.5
        a = sddmm.coo_sddmm(rows, ci, D1, D2); b = sddmm.csr_sddmm(rp, ci, D1, D2); c = sddmm.csr_sddmm(rp, ci, D1, D2, plan=plan)
        same = bool(torch.equal(a.view(torch.int32), b.view(torch.int32)) and torch.equal(a.view(torch.int32), c.view(torch.int32)))
        t_coo = med(lambda: sddmm.coo_sddmm(rows, ci, D1, D2))
        t_csr = med(lambda: sddmm.csr_sddmm(rp, ci, D1, D2))
        t_pl = med(lambda: sddmm.csr_sddmm(rp, ci, D1, D2, plan=plan))
        print("%-16s N=%3d  coo %8.1f us  csr %8.1f us  (csr/coo %.2f)  plan %8.1f us  bits equal: %s" %
              (name, N, t_coo, t_csr, t_csr / t_coo, t_pl, same), flush=True)
