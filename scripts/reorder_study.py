#!/usr/bin/env python3
"""Row-order study for the streaming SpMM kernel (DESIGN.md "Row clustering").

CPU part (runs anywhere): a hierarchical planted-partition stand-in with SHUFFLED vertex ids, multi-level label
propagation as the row order, and an LRU model of the per-XCD L2 (B rows as 512-byte objects) that predicts the
share of B-row gathers served from L2 for a given processing order.
GPU part (--gpu): the same orders applied as a physical row permutation of A, timed through the C ABI.

    python scripts/reorder_study.py [--gpu] [--graph sbm|structureless] [--ncols 128]
"""
import argparse
import os
import sys
import time
from collections import OrderedDict

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def lru_hits(A, order, cap=6000, nxcd=8):
    indptr, ind = A.indptr, A.indices
    deg = np.diff(indptr)[order]
    cum = np.cumsum(deg)
    tot = cum[-1]
    hits = 0
    cuts = [0] + [int(np.searchsorted(cum, tot * (x + 1) // nxcd)) for x in range(nxcd - 1)] + [len(order)]
    for x in range(nxcd):
        od = OrderedDict()
        for r in order[cuts[x]:cuts[x + 1]]:
            for c in ind[indptr[r]:indptr[r + 1]]:
                if c in od:
                    od.move_to_end(c)
                    hits += 1
                else:
                    od[c] = 1
                    if len(od) > cap:
                        od.popitem(last=False)
    return hits / tot


def lp_level(indptr, ind, w, iters=5, seed=0):
    n = indptr.size - 1
    rows = np.repeat(np.arange(n), np.diff(indptr))
    lab = np.arange(n)
    rng = np.random.default_rng(seed)
    for it in range(iters):
        key = rows * n + lab[ind]
        o = np.argsort(key, kind="stable")
        ks, ws = key[o], w[o]
        idx = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
        sums = np.add.reduceat(ws, idx) + rng.random(idx.size) * 0.5
        kr, kl = ks[idx] // n, ks[idx] % n
        o2 = np.lexsort((sums, kr))
        last = np.r_[kr[o2][1:] != kr[o2][:-1], True]
        best_r, best_l = kr[o2][last], kl[o2][last]
        mask = (rng.random(n) < 0.5) if it < iters - 1 else np.ones(n, bool)
        sel = mask[best_r]
        new = lab.copy()
        new[best_r[sel]] = best_l[sel]
        if (new != lab).sum() == 0:
            break
        lab = new
    return np.unique(lab, return_inverse=True)[1]


def multilevel_order(A, levels=8, iters=5):
    indptr, ind = A.indptr.astype(np.int64), A.indices.astype(np.int64)
    w = np.ones(ind.size)
    labs, cur = [], np.arange(A.shape[0])
    for lv in range(levels):
        lab = lp_level(indptr, ind, w, iters, seed=lv)
        nc = lab.max() + 1
        cur = lab[cur]
        labs.append(cur.copy())
        if nc == indptr.size - 1 or nc < 16:
            break
        n = indptr.size - 1
        rows = np.repeat(np.arange(n), np.diff(indptr))
        r, c = lab[rows], lab[ind]
        m = r != c
        C = sp.coo_matrix((w[m], (r[m], c[m])), shape=(nc, nc)).tocsr()
        C.sum_duplicates()
        indptr, ind, w = C.indptr.astype(np.int64), C.indices.astype(np.int64), C.data
        if ind.size == 0:
            break
    return np.lexsort(tuple(labs))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--graph", default="sbm")
    ap.add_argument("--ncols", type=int, default=128)
    ap.add_argument("--iters", type=int, default=200)
    args = ap.parse_args()
    import torch

    import gespmm_amd  # noqa: F401
    from gespmm_amd import _lib, graphs, spmm

    name = "com-amazon-sbm" if args.graph == "sbm" else "com-amazon-like"
    g = graphs.synthetic_graph(name, seed=42, device="cpu")
    M, nnz = g["M"], g["nnz"]
    A = sp.csr_matrix((np.ones(nnz, np.float32), g["colind"].numpy(), g["rowptr"].numpy()), shape=(M, M))
    orders = {"natural": np.arange(M)}
    t0 = time.time()
    orders["multilevel-lp"] = multilevel_order(A)
    print("ordering (numpy prototype): %.1f s" % (time.time() - t0))
    if "truth" in g:
        orders["planted"] = np.argsort(g["truth"].numpy(), kind="stable")
    for k, o in orders.items():
        print("%-14s simulated B-row L2 hits (cap 6000 rows/XCD): %.3f" % (k, lru_hits(A, o)))
    if not args.gpu:
        return
    dev = torch.device("cuda")
    N = args.ncols
    B = ((torch.randint(0, 100, (M, N), device=dev, dtype=torch.int32) - 50).float() / 100)
    val = torch.rand(nnz, device=dev) - 0.5
    ref = None
    for k, o in orders.items():
        Ap = A[o]  # physical row permutation
        rp = torch.from_numpy(Ap.indptr.astype(np.int32)).to(dev)
        ci = torch.from_numpy(Ap.indices.astype(np.int32)).to(dev)
        # values follow their entries
        pos = sp.csr_matrix((np.arange(nnz, dtype=np.float64) + 1, A.indices, A.indptr), shape=(M, M))[o].data
        v = val[torch.from_numpy((pos - 1).astype(np.int64)).to(dev)]
        C = torch.empty((M, N), device=dev)
        for label, cfg in (("auto", None), ("rpw2", {"rows_per_wave": 2, "flags": _lib.FLAG_BATCH_STREAM}),
                           ("rpw4", {"rows_per_wave": 4, "flags": _lib.FLAG_BATCH_STREAM}),
                           ("rpw8", {"rows_per_wave": 8, "flags": _lib.FLAG_BATCH_STREAM}),
                           ("rpw16", {"rows_per_wave": 16, "flags": _lib.FLAG_BATCH_STREAM}),
                           ("rpw32", {"rows_per_wave": 32, "flags": _lib.FLAG_BATCH_STREAM}),
                           ("seg4", {"rows_per_wave": 4, "flags": _lib.FLAG_SEG_STREAM}),
                           ("seg8", {"rows_per_wave": 8, "flags": _lib.FLAG_SEG_STREAM})):
            for _ in range(20):
                spmm.csr_spmm(rp, ci, v, B, cfg=cfg, out=C)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.iters):
                spmm.csr_spmm(rp, ci, v, B, cfg=cfg, out=C)
            e1.record()
            torch.cuda.synchronize()
            print("%-14s %-6s %8.1f us" % (k, label, e0.elapsed_time(e1) / args.iters * 1e3), flush=True)
        # same rows, same bits: un-permute and compare with the natural order
        inv = torch.from_numpy(np.asarray(o)).to(dev)
        full = torch.empty_like(C)
        full[inv] = C
        if ref is None:
            ref = full.clone()
        else:
            assert torch.equal(full.view(torch.int32), ref.view(torch.int32)), "row order changed the bits"


if __name__ == "__main__":
    main()
