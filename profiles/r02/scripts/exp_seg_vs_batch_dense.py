#!/usr/bin/env python3
"""Batch-stream vs segmented-stream kernel where rows are longer than the batch kernel's 64-entry tile (the two lane groups
of a wavefront then take turns instead of gathering side by side)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch

import gespmm_amd  # noqa: F401,E402
from gespmm_amd import graphs, spmm  # noqa: E402

dev = torch.device("cuda")


def timeit(fn, iters=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name, widths in (("reddit-like", (16, 32, 64)), ("products-like", (32, 64, 128)), ("rmat-20", (32, 128))):
    g = graphs.synthetic_graph(name, seed=42, device=dev)
    M, K, nnz = g["M"], g["K"], g["nnz"]
    rp, ci = g["rowptr"], g["colind"]
    val = torch.rand(nnz, device=dev) - 0.5
    for N in widths:
        B = torch.rand((K, N), device=dev)
        C = torch.empty((M, N), device=dev)
        out = []
        for label, flags in (("auto", 0), ("batch strict", 0x120), ("seg strict", 0x180), ("batch+longrows", 0x220)):
            try:
                ms = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, cfg={"flags": flags} if flags else None))
                out.append("%s %.3f ms" % (label, ms))
            except Exception as ex:  # noqa: BLE001
                out.append("%s failed (%s)" % (label, str(ex)[:40]))
        print("%s N=%d: %s" % (name, N, " | ".join(out)), flush=True)
