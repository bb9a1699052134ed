"""Small graphs (launch-latency territory): is the clustered order worth its analysis? plain call, storage-order plan, clustered plan; creation
times (median of 5) — pubmed / cora (real), cit-HepTh-shaped, and the headline graph for scale."""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
import gespmm_amd  # noqa
from gespmm_amd import graphs, spmm
dev = torch.device("cuda")
def med(fn, n=200):
    for _ in range(10): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev)
def cases():
    for name in ("cora", "pubmed"):
        g = graphs.load_mtx_as_csr(os.path.join(ROOT, "tests", "golden", name + ".mtx"))
        yield name, {"M": g["M"], "K": g["K"], "nnz": g["nnz"], "rowptr": torch.from_numpy(g["rowptr"]).to(dev), "colind": torch.from_numpy(g["colind"]).to(dev)}
    for n in ("pubmed-selfloop-like", "cit-hepth-like", "com-amazon-sbm"):
        yield n, graphs.synthetic_graph(n, seed=42, device=dev)
    for sc in (0.05, 0.1, 0.25, 0.5):
        yield "com-amazon-sbm x%.2f" % sc, graphs.synthetic_graph("com-amazon-sbm", seed=42, device=dev, scale=sc)
for name, g in cases():
    rp, ci, K, M, nnz = g["rowptr"], g["colind"], g["K"], g["M"], g["nnz"]
    val = torch.rand(nnz, device=dev) - 0.5
    for N in (32, 128, 512):
        B = torch.rand(K, N, device=dev) - 0.5
        C = torch.empty(M, N, device=dev)
        t_plain = med(lambda: spmm.csr_spmm(rp, ci, val, B, out=C))
        out = []
        for label, kw in (("storage", dict(reorder=False)), ("clustered", dict(reorder=True)), ("auto", dict())):
            ts = []
            for _ in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                p = spmm.SpmmPlan(rp, ci, K, N, values=val, **kw)
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            t = med(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p))
            out.append("%s %.1f us (create %.2f ms, %s)" % (label, t, statistics.median(ts), p.describe().split(" ")[0]))
            del p
        print("%-22s M=%-7d nnz=%-8d N=%-3d plain %.1f us | %s" % (name, M, nnz, N, t_plain, " | ".join(out)), flush=True)
