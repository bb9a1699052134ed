#!/usr/bin/env python3
"""Heavier graphs: reddit-like, products-like(N=64), RMAT scale 22; kernel generations and rows-per-group."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gespmm_amd
from gespmm_amd import _lib as F, graphs, spmm

def time_fn(fn, iters=6, warm=2):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

dev = torch.device("cuda:0")
def run(name, g, Ns):
    rp, ci = g["rowptr"], g["colind"]
    M, K, nnz = g["M"], g["K"], g["nnz"]
    deg = (rp[1:] - rp[:-1])
    print("== %s M=%d K=%d nnz=%d maxdeg=%d" % (name, M, K, nnz, int(deg.max()))); sys.stdout.flush()
    val = torch.rand(nnz, device=dev) - 0.5
    for N in Ns:
        B = torch.rand((K, N), device=dev)
        C = torch.empty((M, N), device=dev)
        line = "  N=%3d:" % N
        for label, variant, cfg in [("auto", -1, None), ("seg g1", 3, dict(rows_per_wave=1)), ("seg g2", 3, dict(rows_per_wave=2)),
                           ("seg g4", 3, dict(rows_per_wave=4)), ("seg g16", 3, dict(rows_per_wave=16)),
                           ("seg g4 u4", 3, dict(rows_per_wave=4, flags=F.FLAG_SHALLOW_UNROLL)),
                           ("bs r8", 3, dict(rows_per_wave=8, flags=F.FLAG_BATCH_STREAM)),
                           ("bs r8 strict", 3, dict(rows_per_wave=8, flags=F.FLAG_BATCH_STREAM | F.FLAG_STRICT_ORDER)),
                           ("bs r8 split", 3, dict(rows_per_wave=8, flags=F.FLAG_BATCH_STREAM | F.FLAG_SPLIT_LONG_ROWS)),
                           ("v4", 4, None), ("v2", 2, None)]:
            us = time_fn(lambda: spmm.csr_spmm(rp, ci, val, B, variant=variant, cfg=cfg, out=C))
            line += " | %s %.0f us (%.2f TF)" % (label, us, 2.0 * nnz * N / us / 1e6)
        print(line); sys.stdout.flush()
        del B, C

t = time.time(); g = graphs.rmat_shard(22, 16, 0, 1, seed=42, device=dev); print("rmat22 gen %.1fs" % (time.time() - t))
run("rmat-22", g, (128, 256)); del g; torch.cuda.empty_cache()
t = time.time(); g = graphs.synthetic_graph("reddit-like", device=dev); print("reddit gen %.1fs" % (time.time() - t))
run("reddit-like", g, (128,)); del g; torch.cuda.empty_cache()
t = time.time(); g = graphs.synthetic_graph("products-like", device=dev); print("products gen %.1fs" % (time.time() - t))
run("products-like", g, (128,)); del g; torch.cuda.empty_cache()
g = graphs.synthetic_graph("cit-hepth-like", device=dev)
run("cit-hepth-like", g, (128,))
g = graphs.synthetic_graph("com-amazon-like", device=dev)
run("com-amazon-like", g, (128,))
