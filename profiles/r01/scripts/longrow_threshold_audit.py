#!/usr/bin/env python3
"""Long-row pass: where should it switch on? RMAT graphs of growing scale, AUTO vs forced split vs strict."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gespmm_amd
from gespmm_amd import _lib as F, graphs, spmm

def time_fn(fn, iters, warm=2):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

dev = torch.device("cuda:0")
for sc in (14, 16, 17, 18, 19, 20):
    g = graphs.rmat_shard(sc, 16, 0, 1, device=dev)
    rp, ci, M, K = g["rowptr"], g["colind"], g["M"], g["K"]
    nnz = ci.numel()
    maxdeg = int((rp[1:] - rp[:-1]).max())
    val = torch.rand(nnz, device=dev)
    for N in (32, 128):
        B = torch.rand(K, N, device=dev); C = torch.empty(M, N, device=dev)
        it = 20
        line = "rmat%d nnz=%d maxdeg=%d N=%d:" % (sc, nnz, maxdeg, N)
        for label, flags in (("auto", 0), ("split", F.FLAG_SPLIT_LONG_ROWS), ("strict", F.FLAG_STRICT_ORDER)):
            line += " %s %.1f |" % (label, time_fn(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, cfg=dict(flags=flags)), it))
        print(line); sys.stdout.flush()
