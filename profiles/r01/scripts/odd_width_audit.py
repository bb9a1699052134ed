#!/usr/bin/env python3
"""Odd widths (class logits: N=41 reddit, N=3 pubmed, N=47 products): AUTO vs explicit geometries."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gespmm_amd
from gespmm_amd import _lib as F, graphs, spmm

def time_fn(fn, iters, warm=1):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

dev = torch.device("cuda:0")
for name in sys.argv[1].split(","):
    g = graphs.synthetic_graph(name, device=dev)
    M, K, nnz = g["M"], g["K"], g["nnz"]
    rp, ci = g["rowptr"], g["colind"]
    val = torch.rand(nnz, device=dev)
    it = 3 if nnz > 2e7 else 50
    for N in (32, 41, 47, 64):
        B = torch.rand(K, N, device=dev); C = torch.empty(M, N, device=dev)
        line = "%-16s N=%2d:" % (name, N)
        for label, cfg in (("auto", None), ("W64", dict(group=64)), ("W32", dict(group=32)), ("W16", dict(group=16)),
                           ("W32 noslab", dict(group=32, flags=F.FLAG_NO_SLAB_BLOCKED)), ("W64 noslab", dict(group=64, flags=F.FLAG_NO_SLAB_BLOCKED))):
            line += " %s %.0f |" % (label, time_fn(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, cfg=cfg), it))
        print(line); sys.stdout.flush()
