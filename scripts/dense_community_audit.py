import sys, statistics, torch
sys.path.insert(0, ".")
import gespmm_amd
from gespmm_amd import graphs, spmm
def timed(fn, reps=7):
    for _ in range(2): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev)
M, nnz = 232965, 114615892
cases = [("reddit-sbm(290 comm, intra 330)", graphs.community_csr(M, nnz, 290, 16, 330.0, 0.6, 1.5, 1.55, 42, "cuda")[:2]),
         ("reddit-sbm(1200 comm, intra 150)", graphs.community_csr(M, nnz, 1200, 16, 150.0, 0.6, 1.5, 1.55, 42, "cuda")[:2])]
g = graphs.synthetic_graph("reddit-like", seed=42, device="cuda")
cases.append(("reddit-like (structureless)", (g["rowptr"], g["colind"])))
for name, (rp, ci) in cases:
    val = torch.rand(int(ci.numel()), device="cuda") - 0.5
    for N in (64, 128, 256):
        B = torch.rand(M, N, device="cuda") - 0.5
        C = torch.empty(M, N, device="cuda")
        p0 = spmm.SpmmPlan(rp, ci, M, N, values=val)
        t0 = timed(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p0))
        p1 = spmm.SpmmPlan(rp, ci, M, N, values=val, reorder=True, flags=0x800)
        t1 = timed(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p1))
        d = p1.describe()
        print("%-34s N=%3d  AUTO (%s) %8.1f us | forced clustering %8.1f us x%.2f | %s | %s" % (name, N, p0.describe().split("kernel=")[1].split()[0], t0, t1, t0 / t1, d.split("l2_model=")[1].split()[0] if "l2_model=" in d else "-", d.split("kernel=")[1].split()[0]), flush=True)
        del p0, p1
