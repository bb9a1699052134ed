import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import gespmm_amd
from gespmm_amd import graphs, spmm
dev = torch.device("cuda")
for name in ("pubmed-like", "cit-hepth-like", "com-amazon-like"):
    g = graphs.synthetic_graph(name, seed=42, device=dev)
    rp, ci, K = g["rowptr"], g["colind"], g["K"]
    val = torch.rand(g["nnz"], device=dev)
    for N in (128,):
        ts = []
        for rep in range(5):
            torch.cuda.synchronize(); t = time.perf_counter()
            p = spmm.SpmmPlan(rp, ci, K, N, values=val)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        print(name, N, "plan create ms:", " ".join("%.1f" % x for x in ts), "|", p.describe()[:150])
