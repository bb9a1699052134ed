#!/usr/bin/env python3
"""Would a row schedule pay? Time the standard SpMM on the row-permuted matrix P*A (same B): BFS order
(rows sharing a neighbour become adjacent), against the natural order. C comes out row-permuted, which a
planned path would undo for free by storing each row at its natural position."""
import os, sys
from collections import deque
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import gespmm_amd
from gespmm_amd import _lib as F, graphs, spmm

def time_fn(fn, iters=200, warm=20):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

def bfs_order(rp, ci, M):
    seen = np.zeros(M, bool); order = []
    deg = np.diff(rp)
    for s in np.argsort(-deg):
        if seen[s]: continue
        seen[s] = True; q = deque([s])
        while q:
            u = q.popleft(); order.append(u)
            for v in ci[rp[u]:rp[u + 1]]:
                if not seen[v]:
                    seen[v] = True; q.append(v)
    return np.array(order)

def permute_rows(rp, ci, order):
    deg = np.diff(rp)[order]
    nrp = np.zeros(len(order) + 1, np.int64); nrp[1:] = np.cumsum(deg)
    idx = np.concatenate([np.arange(rp[r], rp[r + 1]) for r in order]) if len(order) else np.zeros(0, np.int64)
    return nrp.astype(np.int32), ci[idx].astype(np.int32)

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "com-amazon-like"
g = graphs.synthetic_graph(name)
M, K = g["M"], g["K"]
rp = g["rowptr"].numpy().astype(np.int64); ci = g["colind"].numpy().astype(np.int64)
MODE = os.environ.get("ORDER", "bfs")
if MODE == "bfs":
    order = bfs_order(rp, ci, M)
elif MODE == "degdesc":
    order = np.argsort(-np.diff(rp), kind="stable")
elif MODE == "heavyfirst":  # rows above 64 entries first (in natural order), the rest in natural order
    d = np.diff(rp)
    order = np.concatenate([np.nonzero(d > 64)[0], np.nonzero(d <= 64)[0]])
rp2, ci2 = permute_rows(rp, ci, order)
for N in (128, 32, 512):
    B = torch.rand(K, N, device=dev); C = torch.empty(M, N, device=dev)
    line = "%s N=%d:" % (name, N)
    for label, (a, b) in (("natural", (g["rowptr"].to(dev), g["colind"].to(dev))), (MODE + " rows", (torch.from_numpy(rp2).to(dev), torch.from_numpy(ci2).to(dev)))):
        val = torch.rand(b.numel(), device=dev)
        for kname, cfg in (("auto", None), ("noremap", dict(flags=F.FLAG_NO_XCD_REMAP)), ("batch r8", dict(rows_per_wave=8, flags=F.FLAG_BATCH_STREAM))):
            line += " %s/%s %.1f |" % (label, kname, time_fn(lambda: spmm.csr_spmm(a, b, val, B, out=C, cfg=cfg)))
    print(line); sys.stdout.flush()
