#!/usr/bin/env python3
"""System-scope (sc1) C stores vs the default store flavour: plain call and clustered plans on the bench graphs, slab path on the
reddit-shaped graph (GESPMM_EXP_SC1=1 switches the slab kernel)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch

import gespmm_amd  # noqa: F401,E402
from gespmm_amd import graphs, spmm  # noqa: E402

dev = torch.device("cuda")
SC1 = 0x8000


def timeit(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


which = sys.argv[1] if len(sys.argv) > 1 else "all"
names = ("com-amazon-sbm", "com-amazon-like", "products-sbm", "pubmed-like") if which == "all" else (which,)
for name in names:
    g = graphs.synthetic_graph(name, seed=42, device=dev)
    M, K, nnz = g["M"], g["K"], g["nnz"]
    rp, ci = g["rowptr"], g["colind"]
    val = torch.rand(nnz, device=dev) - 0.5
    iters = 200 if nnz < 5e6 else 10
    for N in (32, 128, 512) if nnz < 5e7 else (32, 128):
        B = torch.rand((K, N), device=dev)
        C = torch.empty((M, N), device=dev)
        row = []
        for lab, fl in (("default", 0), ("sc1", SC1)):
            us = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, cfg={"flags": fl} if fl else None), iters)
            row.append("plain %s %.1f" % (lab, us))
        if name != "reddit-like":
            for lab, fl in (("default", 0), ("sc1", SC1)):
                plan = spmm.SpmmPlan(rp, ci, K, N, values=val, flags=fl)
                us = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan), iters)
                row.append("plan %s %.1f" % (lab, us))
        print("%-16s N=%-4d %s" % (name, N, " | ".join(row)), flush=True)
