#!/usr/bin/env python3
"""Segmented-stream kernel on clustered plans: lane-group task size x unroll depth (com-amazon stand-ins, N = 128)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch

import gespmm_amd  # noqa: F401,E402
from gespmm_amd import graphs, spmm  # noqa: E402

dev = torch.device("cuda")


def timeit(fn, iters=200):
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


for name, N in (("com-amazon-sbm", 128), ("com-amazon-like", 128), ("com-amazon-like", 256), ("com-amazon-like", 512), ("com-amazon-sbm", 512)):
    g = graphs.synthetic_graph(name, seed=42, device=dev)
    M, K, nnz = g["M"], g["K"], g["nnz"]
    rp, ci = g["rowptr"], g["colind"]
    val = torch.rand(nnz, device=dev) - 0.5
    B = torch.rand((K, N), device=dev)
    C = torch.empty((M, N), device=dev)
    for kernel in ("stream", "seg-stream"):
        for flags, lab in ((0x20000, "U8"),):
            row = []
            for te in ((0, 64, 96, 128) if kernel == "stream" else (0, 8, 16, 24, 32, 48, 64)):
                plan = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True, task_entries=te, kernel=kernel, flags=flags)
                us = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan))
                row.append("%d:%.1f" % (te, us))
            print("%s N=%d %-10s %s  %s" % (name, N, kernel, lab, " ".join(row)), flush=True)
