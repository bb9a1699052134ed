#!/usr/bin/env python3
"""Slab size sweep of the slab-blocked path on reddit-like."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gespmm_amd
from gespmm_amd import _lib as F, graphs, spmm

def time_fn(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

dev = torch.device("cuda:0")
g = graphs.synthetic_graph("reddit-like", device=dev)
M, K, nnz = g["M"], g["K"], g["nnz"]
rp, ci = g["rowptr"], g["colind"]
val = torch.rand(nnz, device=dev)
for N in (128, 64, 32, 256, 512):
    B = torch.rand(K, N, device=dev); C = torch.empty(M, N, device=dev)
    base = time_fn(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, cfg=dict(flags=F.FLAG_NO_SLAB_BLOCKED)))
    line = "N=%3d streaming %.0f us |" % (N, base)
    for mb in (2, 3, 4, 6, 8, 12, 16, 24):
        row_bytes = min(N, 256) * 4 if N >= 64 else N * 4
        slab_rows = (mb << 20) // row_bytes
        us = time_fn(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, cfg=dict(slab_rows=slab_rows, flags=F.FLAG_SLAB_BLOCKED)))
        line += " %dMB(%d rows,%d slabs) %.0f |" % (mb, slab_rows, (K + slab_rows - 1) // slab_rows, us)
    print(line); sys.stdout.flush()
    del B, C
