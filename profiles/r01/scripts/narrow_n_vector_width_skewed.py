#!/usr/bin/env python3
"""N <= 64 on skewed graphs: which vector width? (auto picks V=1 for N <= 64)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gespmm_amd
from gespmm_amd import _lib as F, graphs, spmm

def time_fn(fn, iters, warm):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

dev = torch.device("cuda:0")
names = sys.argv[1].split(",")
for name in names:
    if name.startswith("rmat"):
        g = graphs.rmat_shard(int(name[4:]), 16, 0, 1, device=dev)
    else:
        g = graphs.synthetic_graph(name, device=dev)
    rp, ci, M, K = g["rowptr"], g["colind"], g["M"], g["K"]
    nnz = ci.numel()
    val = torch.rand(nnz, device=dev)
    big = nnz > 2e7
    for N in (16, 32, 64):
        B = torch.rand(K, N, device=dev); C = torch.empty(M, N, device=dev)
        line = "%-16s N=%2d:" % (name, N)
        for label, variant, cfg in (("auto", -1, None), ("v1", 1, None), ("v2", 2, None), ("v3", 3, None),
                                    ("v1 nosplit", 1, dict(flags=F.FLAG_STRICT_ORDER)), ("v3 nosplit", 3, dict(flags=F.FLAG_STRICT_ORDER)),
                                    ("v1 W32", 1, dict(group=32)), ("v1 W16", 1, dict(group=16))):
            us = time_fn(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, variant=variant, cfg=cfg), 5 if big else 50, 1 if big else 5)
            line += " %s %.0f |" % (label, us)
        print(line); sys.stdout.flush()
        del B, C
