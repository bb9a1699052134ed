#!/usr/bin/env python3
"""Slab-blocked path on reddit-like: lane-group width (= N strips bound to XCDs) x per-XCD slab footprint."""
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
    ref = None
    print("N=%d auto: %.0f us" % (N, time_fn(lambda: spmm.csr_spmm(rp, ci, val, B, out=C)))); sys.stdout.flush()
    ref = C.clone()
    for group in (32,):
        if group * 4 > N: continue
        strip_bytes = group * 16
        line = "  group %2d (strip %3d B, %d strips):" % (group, strip_bytes, N * 4 // strip_bytes)
        for mb in (2, 3, 4, 6, 8, 12):
            slab_rows = int(mb * (1 << 20)) // strip_bytes
            cfg = dict(vec=4, strips=1, group=group, slab_rows=slab_rows, flags=F.FLAG_SLAB_BLOCKED)
            us = time_fn(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, cfg=cfg))
            ok = torch.equal(C, ref)
            line += " %.1fMB/%dsl %.0f%s |" % (mb, (K + slab_rows - 1) // slab_rows, us, "" if ok else "(!)")
        print(line); sys.stdout.flush()
    del B, C
