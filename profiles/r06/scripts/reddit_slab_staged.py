"""Feasibility of COLUMN-SLAB staging on the dense community graph (round-5 review, item 6), with the kernels as they are: the matrix is cut into
P column ranges, each range is its own matrix (same rows) with its own forced staged-rows plan, and the P products are timed one by one. A
multi-phase kernel (accumulators carried through C between phases: + 2 x M x N x 4 bytes per extra phase) could not be faster than the sum
printed here + that traffic.   python profiles/r06/scripts/reddit_slab_staged.py [graph] [N]"""
import statistics, sys
import torch
sys.path.insert(0, ".")
import gespmm_amd
from gespmm_amd import graphs, spmm

name = sys.argv[1] if len(sys.argv) > 1 else "reddit-sbm"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128


def timed(fn, reps):
    for _ in range(2): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev)


g = graphs.synthetic_graph(name, seed=42, device="cuda")
rp, ci, M, K, nnz = g["rowptr"], g["colind"], g["M"], g["K"], g["nnz"]
val = torch.rand(nnz, device="cuda") - 0.5
B = torch.rand(K, N, device="cuda") - 0.5
C = torch.empty(M, N, device="cuda")
rows = torch.repeat_interleave(torch.arange(M, device="cuda"), (rp[1:] - rp[:-1]).long())
for kern in ("auto", "staged", "seg-stream"):
    plan = spmm.SpmmPlan(rp, ci, K, N, values=val, kernel=kern)
    t = timed(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan), 7)
    print("%s N=%d whole matrix kernel=%-10s %9.1f us | %s" % (name, N, kern, t, plan.describe().split("|")[-1].strip()[:150]), flush=True)
    del plan
for P in (2, 3, 5, 8, 12):
    total, notes = 0.0, []
    for p in range(P):
        lo, hi = (K * p) // P, (K * (p + 1)) // P
        m = (ci >= lo) & (ci < hi)
        cip, vp = ci[m].contiguous(), val[m].contiguous()
        rpp = torch.zeros(M + 1, dtype=torch.int64, device="cuda")
        rpp[1:] = torch.cumsum(torch.bincount(rows[m], minlength=M), 0)
        rpp = rpp.to(torch.int32)
        for kern in ("staged",):
            plan = spmm.SpmmPlan(rpp, cip, K, N, values=vp, kernel=kern, reorder=True)
            t = timed(lambda: spmm.csr_spmm(rpp, cip, vp, B, out=C, plan=plan), 5)
            d = plan.describe()
            se = d.split("staged_entries=")[1].split()[0] if "staged_entries=" in d else "-"
            notes.append("%.0f us (staged %s)" % (t, se))
            total += t
            del plan
    extra = 2.0 * M * N * 4 * (P - 1) / 6.5e6
    print("%s N=%d P=%2d slabs: sum %9.1f us, + %.0f us of C traffic at 6.5 TB/s = %9.1f | %s" % (name, N, P, total, extra, total + extra, ", ".join(notes)[:400]), flush=True)
