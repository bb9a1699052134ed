"""The reference's own kernels (oracle/_ref/libref_kernels.so = spmm_test.cu compiled by hipcc for gfx950, a baseline leg) against this
library on the BASELINE graph shapes, same operands, same GPU; results compared bit for bit.
    python scripts/reference_kernels_compare.py"""
import os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gespmm_amd  # noqa: E402
from gespmm_amd import graphs, spmm  # noqa: E402
import ref_py  # noqa: E402


def med(fn, n):
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev)


cases = [("cit-hepth-like", 32, 200), ("pubmed-like", 128, 200), ("com-amazon-sbm", 128, 200), ("com-amazon-like", 128, 200),
         ("com-amazon-sbm", 512, 50), ("reddit-like", 128, 5), ("products-sbm", 128, 5), ("products-like", 128, 5)]
print("%-16s %4s | %-34s | %-22s | %-22s | bits" % ("graph", "N", "reference kernels: method 2 (timed by the reference) / best of 0-4", "this library, plain", "this library, plan"))
for name, N, reps in cases:
    g = graphs.synthetic_graph(name, seed=42, device="cuda")
    rp, ci, M, K, nnz = g["rowptr"], g["colind"], g["M"], g["K"], g["nnz"]
    if K * N >= (1 << 31):
        print("%-16s %4d | skipped: the reference pre-multiplies column indices in int32 (K*N >= 2^31)" % (name, N)); continue
    val = torch.rand(nnz, device="cuda") - 0.5
    B = (torch.randint(0, 100, (K, N), device="cuda", dtype=torch.int32) - 50).float() / 100
    C = torch.empty(M, N, device="cuda")
    Cr = torch.empty(M, N, device="cuda")
    t_ref = {}
    for method in range(5):
        t_ref[method] = med(lambda: ref_py.spmm_wrapper(method, 8, rp, ci, val, B, out=Cr, sync=False), reps)
    ref_py.spmm_wrapper(2, 8, rp, ci, val, B, out=Cr)
    t_plain = med(lambda: spmm.csr_spmm(rp, ci, val, B, out=C), reps)
    same = bool(torch.equal(C.view(torch.int32), Cr.view(torch.int32)))
    plan = spmm.SpmmPlan(rp, ci, K, N, values=val)
    t_plan = med(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan), reps)
    same_p = bool(torch.equal(C.view(torch.int32), Cr.view(torch.int32)))
    best = min(t_ref, key=t_ref.get)
    fl = 2.0 * nnz * N
    print("%-16s %4d | m2 %9.1f us %7.1f GF/s; best m%d %9.1f us | %9.1f us %7.1f GF/s x%.2f | %9.1f us %7.1f GF/s x%.2f | %s %s" %
          (name, N, t_ref[2], fl / t_ref[2] / 1e3, best, t_ref[best], t_plain, fl / t_plain / 1e3, t_ref[2] / t_plain,
           t_plan, fl / t_plan / 1e3, t_ref[2] / t_plan, "equal" if same else "DIFFER(long rows: tolerance class)" , "equal" if same_p else "differ"), flush=True)
    del g, val, B, C, Cr, plan
    torch.cuda.empty_cache()
