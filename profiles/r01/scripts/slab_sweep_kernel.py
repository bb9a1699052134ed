#!/usr/bin/env python3
"""Slab-sweep kernel on reddit-like: slab size and rows-per-group sweep vs the streaming kernel."""
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
for name, Ns in (("reddit-like", (128, 64, 32, 256)), ("products-like", (128,)), ("com-amazon-like", (128,))):
    g = graphs.synthetic_graph(name, device=dev)
    M, K, nnz = g["M"], g["K"], g["nnz"]
    rp, ci = g["rowptr"], g["colind"]
    val = torch.rand(nnz, device=dev)
    for N in Ns:
        B = torch.rand(K, N, device=dev); C = torch.empty(M, N, device=dev)
        base = time_fn(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, cfg=dict(flags=F.FLAG_NO_SLAB_BLOCKED)))
        auto = time_fn(lambda: spmm.csr_spmm(rp, ci, val, B, out=C))
        ref = spmm.csr_spmm(rp, ci, val, B, cfg=dict(flags=F.FLAG_NO_SLAB_BLOCKED))
        chk = torch.equal(ref, spmm.csr_spmm(rp, ci, val, B, cfg=dict(flags=F.FLAG_SLAB_BLOCKED)))
        print("== %s N=%d: streaming %.0f us (%.2f TF) ; auto %.0f us (%.2f TF) ; slab-sweep bit-equal: %s" %
              (name, N, base, 2.0 * nnz * N / base / 1e6, auto, 2.0 * nnz * N / auto / 1e6, chk)); sys.stdout.flush()
        if name != "reddit-like":
            continue
        for slab_rows in (2048, 4096, 6144, 8192, 12288):
            line = "   slab_rows %5d:" % slab_rows
            for R in (4, 8, 16):
                us = time_fn(lambda: spmm.csr_spmm(rp, ci, val, B, out=C,
                                                   cfg=dict(slab_rows=slab_rows, rows_per_wave=R, flags=F.FLAG_SLAB_BLOCKED)))
                line += " | R=%d %.0f us (%.2f TF)" % (R, us, 2.0 * nnz * N / us / 1e6)
            print(line); sys.stdout.flush()
        del B, C
    del g
