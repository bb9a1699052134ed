#!/usr/bin/env python3
"""Slab-blocked path on reddit-like: rows per lane group (task size) x slab size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gespmm_amd
from gespmm_amd import _lib as F, graphs, spmm

def time_fn(fn, iters=4, warm=1):
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
Ns = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [128]
for N in Ns:
    B = torch.rand(K, N, device=dev); C = torch.empty(M, N, device=dev)
    print("N=%d auto: %.0f us" % (N, time_fn(lambda: spmm.csr_spmm(rp, ci, val, B, out=C)))); sys.stdout.flush()
    for mb in (3, 4, 6):
        line = "  slab %d MB:" % mb
        for R in (1, 2, 4, 8):
            tile_bytes = min(N * 4, 512)
            cfg = dict(rows_per_wave=R, slab_rows=(mb << 20) // tile_bytes, flags=F.FLAG_SLAB_BLOCKED)
            line += " R%d %.0f |" % (R, time_fn(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, cfg=cfg)))
        print(line); sys.stdout.flush()
    del B, C
