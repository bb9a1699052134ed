#!/usr/bin/env python3
"""Clustered plans: batch-stream vs segmented-stream kernel across feature widths (default task sizes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch

import gespmm_amd  # noqa: F401,E402
from gespmm_amd import graphs, spmm  # noqa: E402

dev = torch.device("cuda")


def timeit(fn, iters=100):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name in ("com-amazon-sbm", "com-amazon-like", "pubmed", "cit-hepth-like", "products-sbm"):
    try:
        g = graphs.synthetic_graph(name, seed=42, device=dev)
    except Exception:  # noqa: BLE001
        g = graphs.synthetic_graph(name + "-like", seed=42, device=dev)
    M, K, nnz = g["M"], g["K"], g["nnz"]
    rp, ci = g["rowptr"], g["colind"]
    val = torch.rand(nnz, device=dev) - 0.5
    for N in (16, 32, 64, 128, 256, 512):
        if name == "products-sbm" and N > 128:
            continue
        B = torch.rand((K, N), device=dev)
        C = torch.empty((M, N), device=dev)
        row = []
        for kernel in ("auto", "stream", "seg-stream"):
            plan = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True, kernel=kernel)
            us = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan), 100 if nnz < 5e7 else 10)
            row.append("%s %.1f" % (kernel, us))
        print("%-16s N=%-4d %s" % (name, N, " | ".join(row)), flush=True)
