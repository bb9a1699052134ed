"""Clustering depth under the staged-rows kernel (products-shaped communities, quarter and full size) and on the headline graph: 3 / 4 / 5 / 6
levels interleaved in one process, two rounds."""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
import gespmm_amd  # noqa
from gespmm_amd import graphs, spmm
dev = torch.device("cuda")
def med(fn, n):
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev)
for name, scale, widths in (("products-sbm", 0.25, (32, 128, 256)), ("products-sbm", 1.0, (128, 256)), ("com-amazon-sbm", 1.0, (128, 256))):
    g = graphs.synthetic_graph(name, seed=42, device=dev, scale=scale)
    rp, ci, K, M, nnz = g["rowptr"], g["colind"], g["K"], g["M"], g["nnz"]
    val = torch.rand(nnz, device=dev) - 0.5
    for N in widths:
        B = torch.rand(K, N, device=dev) - 0.5
        C = torch.empty(M, N, device=dev)
        plans = {}
        for lv in (3, 4, 5, 6):
            os.environ["GESPMM_CLUSTER_LEVELS"] = str(lv)
            plans[lv] = spmm.SpmmPlan(rp, ci, K, N, values=val)
        del os.environ["GESPMM_CLUSTER_LEVELS"]
        n = 60 if nnz < 1e7 else (20 if nnz < 5e7 else 8)
        r = [{lv: med(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p), n) for lv, p in plans.items()} for _ in range(2)]
        kern = "staged" if "staged-rows" in plans[6].describe() else ("seg" if "segmented" in plans[6].describe() else "batch")
        print("%-14s x%.2f N=%-3d %-6s " % (name, scale, N, kern) + " | ".join("L%d %.1f %.1f" % (lv, r[0][lv], r[1][lv]) for lv in plans), flush=True)
        del plans
