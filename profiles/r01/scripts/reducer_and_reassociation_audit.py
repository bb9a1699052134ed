#!/usr/bin/env python3
"""Audit of the reducers and of the opt-in re-association rule: max vs sum; AUTO vs AUTO+ALLOW_REASSOCIATION
(parallel-reduction variant for N <= 16 on rows of >= 32 entries) vs variant 5 forced."""
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
for name in ("com-amazon-like", "products-like", "reddit-like", "rmat20"):
    g = graphs.rmat_shard(20, 16, 0, 1, device=dev) if name == "rmat20" else graphs.synthetic_graph(name, device=dev)
    rp, ci, M, K = g["rowptr"], g["colind"], g["M"], g["K"]
    val = torch.rand(ci.numel(), device=dev)
    it = 3 if ci.numel() > 2e7 else 50
    for N in (128,):
        B = torch.rand(K, N, device=dev)
        print("%-16s N=%3d  sum(unweighted) %9.1f  max %9.1f" % (name, N,
              time_fn(lambda: spmm.csr_spmm_no_edge_value(rp, ci, B), it), time_fn(lambda: spmm.csr_spmm_max(rp, ci, B), it)))
    for N in (1, 4, 8, 16):
        B = torch.rand(K, N, device=dev); C = torch.empty(M, N, device=dev)
        a = time_fn(lambda: spmm.csr_spmm(rp, ci, val, B, out=C), it)
        b = time_fn(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, cfg=dict(flags=F.FLAG_ALLOW_REASSOCIATION)), it)
        c = time_fn(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, variant=5), it)
        print("%-16s N=%3d  auto %9.1f  auto+reassoc %9.1f  variant5 %9.1f%s" % (name, N, a, b, c,
              "  <-- reassoc rule misses" if min(b, a) > 1.15 * min(a, c) else ""))
    sys.stdout.flush()
