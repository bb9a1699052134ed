#!/usr/bin/env python3
"""Slab-resident kernel (accumulators in LDS) vs the slab-blocked path on reddit-like."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch

import gespmm_amd  # noqa: F401,E402
from gespmm_amd import graphs, spmm  # noqa: E402

dev = torch.device("cuda")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
g = graphs.synthetic_graph("reddit-like", seed=42, device=dev)
M, K, nnz = g["M"], g["K"], g["nnz"]
rp, ci = g["rowptr"], g["colind"]
val = torch.rand(nnz, device=dev) - 0.5
B = ((torch.randint(0, 100, (K, N), device=dev, dtype=torch.int32) - 50).float() / 100)
C = torch.empty((M, N), device=dev)


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


ms = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C))
ref = C.clone()
print("AUTO (slab-blocked)            %.3f ms" % ms, flush=True)
for hub in (0, 4096):
    os.environ["GESPMM_SLABRES_HUB"] = str(hub)
    for k in (0, 4):
        os.environ["GESPMM_SLABRES_K"] = str(k)
        for sr in (2048, 3072, 4096, 6144, 8192, 12288):
            C.zero_()
            cfg = {"flags": 0x400 | 0x80000, "slab_rows": sr}
            ms = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, cfg=cfg))
            bad = (C.view(torch.int32) != ref.view(torch.int32)).any(dim=1).sum().item()
            print("resident hub_thr=%-5d k=%d slab_rows=%-6d %.3f ms   rows with different bits: %d" % (hub, k, sr, ms, bad), flush=True)
