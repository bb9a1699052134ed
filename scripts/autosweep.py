#!/usr/bin/env python3
"""AUTO variant across the graph families and N: one table, used before/after heuristic changes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gespmm_amd
from gespmm_amd import graphs, spmm

def time_fn(fn, iters, warm):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

dev = torch.device("cuda:0")
NS = (16, 32, 64, 128, 256, 512)
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["pubmed-like", "cit-hepth-like", "com-amazon-like", "products-like", "reddit-like"]
print("%-18s" % "graph" + "".join("%12s" % ("N=%d" % n) for n in NS))
for name in names:
    if name.startswith("rmat"):
        sc = int(name[4:])
        g = graphs.rmat_shard(sc, 16, 0, 1, device=dev)
        rp, ci, M, K = g["rowptr"], g["colind"], g["M"], g["K"]
    else:
        g = graphs.synthetic_graph(name, device=dev)
        rp, ci, M, K = g["rowptr"], g["colind"], g["M"], g["K"]
    nnz = ci.numel()
    val = torch.rand(nnz, device=dev)
    big = nnz > 2e7
    line = "%-18s" % name
    for N in NS:
        B = torch.rand(K, N, device=dev); C = torch.empty(M, N, device=dev)
        us = time_fn(lambda: spmm.csr_spmm(rp, ci, val, B, out=C), 5 if big else 200, 1 if big else 20)
        line += "%12.1f" % us
        del B, C
    print(line); sys.stdout.flush()
