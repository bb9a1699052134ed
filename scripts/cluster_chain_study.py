#!/usr/bin/env python3
"""Host-only study (round 6): the plan's order places its coarsest clusters in the order their labels happen to compact to. Does placing
strongly connected clusters NEXT to each other (greedy heaviest-edge chaining of the cluster graph) raise the modelled L2 hit rate?
    python scripts/cluster_chain_study.py [graph[@scale]] ..."""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gespmm_amd import _lib, graphs  # noqa: E402

lib = _lib.lib
lib.gespmm_cluster_rows_study.restype = ctypes.c_int
lib.gespmm_cluster_rows_study.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int64] * 2 + [ctypes.c_int32] * 2 + [ctypes.c_void_p] * 3
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)


def model(rp, ci, M, K, perm, window):
    return lib.gespmm_simulate_l2_hits(P(rp), P(ci), M, K, P(perm) if perm is not None else None, 8, window)


def chain(top, rp, ci, M, min_w=2):
    """Order of the coarsest clusters: edges of the cluster graph (square matrix: the cluster of column c is the cluster of row c) taken
    heaviest first; an edge joins two path ENDS (union-find, degree <= 2); the paths are laid end to end in the order of their
    smallest cluster id."""
    rows = np.repeat(np.arange(M, dtype=np.int64), np.diff(rp))
    a, b = top[rows].astype(np.int64), top[ci].astype(np.int64)
    keep = a != b
    lo, hi = np.minimum(a[keep], b[keep]), np.maximum(a[keep], b[keep])
    nc = int(top.max()) + 1
    key, w = np.unique(lo * nc + hi, return_counts=True)
    sel = w >= min_w
    key, w = key[sel], w[sel]
    order = np.lexsort((key, -w))
    parent = np.arange(nc)
    deg = np.zeros(nc, dtype=np.int8)
    nbr = [[] for _ in range(nc)]

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for e in order:
        u, v = int(key[e] // nc), int(key[e] % nc)
        if deg[u] >= 2 or deg[v] >= 2:
            continue
        ru, rv = find(u), find(v)
        if ru == rv:
            continue
        parent[ru] = rv
        deg[u] += 1
        deg[v] += 1
        nbr[u].append(v)
        nbr[v].append(u)
    seen = np.zeros(nc, dtype=bool)
    out = []
    for s in range(nc):
        if seen[s] or deg[s] == 2:
            continue
        prev, cur = -1, s
        while cur != -1 and not seen[cur]:
            seen[cur] = True
            out.append(cur)
            nxt = [x for x in nbr[cur] if x != prev]
            prev, cur = cur, (nxt[0] if nxt else -1)
    for s in range(nc):
        if not seen[s]:
            out.append(s)
    rank = np.empty(nc, dtype=np.int64)
    rank[np.array(out)] = np.arange(nc)
    return rank, int(sel.sum())


for name in sys.argv[1:] or ["com-amazon-sbm", "products-sbm@0.25", "com-amazon-like"]:
    nm, _, sc = name.partition("@")
    g = graphs.synthetic_graph(nm, seed=42, device="cpu", **({"scale": float(sc)} if sc else {}))
    M, K = g["M"], g["K"]
    rp, ci = g["rowptr"].numpy().astype(np.int32), g["colind"].numpy().astype(np.int32)
    for N in (128,):
        window = (3 << 20) // (4 * N)
        for levels, sweeps in ((3, 5), (6, 5)):
            perm = np.empty(M, dtype=np.int32)
            top = np.empty(M, dtype=np.int32)
            cl = np.zeros(16, dtype=np.int32)
            t0 = time.time()
            nl = lib.gespmm_cluster_rows_study(P(rp), P(ci), M, K, levels, sweeps, P(perm), P(top), P(cl))
            t_cl = time.time() - t0
            base = model(rp, ci, M, K, perm, window)
            t0 = time.time()
            rank, nedges = chain(top, rp, ci, M)
            t_ch = time.time() - t0
            pos = np.empty(M, dtype=np.int64)
            pos[perm] = np.arange(M)  # position of every row in the plan's order (keeps the inner order of a cluster)
            perm2 = np.lexsort((pos, rank[top])).astype(np.int32)
            chained = model(rp, ci, M, K, perm2, window)
            line = "%-18s N=%d levels=%d clusters %s: storage %.3f | plan order %.3f | coarsest clusters chained (%d edges of weight >= 2) %.3f" % (
                name, N, nl, ">".join(str(x) for x in cl[:nl]), model(rp, ci, M, K, None, window), base, nedges, chained)
            if "truth_group" in g:
                tg = g["truth_group"].numpy()
                perm3 = np.lexsort((pos, tg)).astype(np.int32)
                line += " | plan order re-sorted by the TRUE group %.3f" % model(rp, ci, M, K, perm3, window)
            print(line + "  (clustering %.2f s, chaining %.2f s in numpy)" % (t_cl, t_ch), flush=True)
