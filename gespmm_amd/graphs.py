"""Graph inputs for the SpMM path: MatrixMarket files through the C ABI, and seeded
synthetic stand-ins for the SNAP / DGL / OGB graphs BASELINE.json names (there is no
network on either box).

Loader side (host memory, numpy):
    read_mtx(path)                       <- readMtx<float>()          util/util.hpp:286-333
    coo_to_csr(nrows, ncols, row, col)   <- inline COO->CSR           spmm_test.cu:557-581
    load_mtx_as_csr(path)                =  both, values forced to 1  spmm_test.cu:574

Synthetic side (torch, runs on whatever device the generator is given — the big
graphs are built on the GPU). Every generator is a pure function of (name, seed):
    synthetic_graph("com-amazon-like" | "cit-hepth-like" | "reddit-like" |
                    "products-like" | "pubmed-selfloop-like", ...)
    synthetic_graph("com-amazon-sbm" | "products-sbm", ...)   planted communities, vertex ids SHUFFLED
Node ids carry NO locality by default (ids are scrambled by an affine bijection):
the reference's own spreadsheet numbers for com-Amazon/pubmed are consistent with
every B-row gather missing a 2.75 MB L2 (DESIGN.md "Synthetic graphs"), so the
stand-ins do not assume an id ordering that would flatter the caches. ``locality``
> 0 re-introduces a banded community structure for sensitivity runs.
"""
import ctypes
import math

import numpy as np
import torch

from ._lib import Coo, check, lib

# name -> (M, nnz, symmetric, gamma)   sizes from SURVEY.md §8 config table
SPECS = {
    "cit-hepth-like": (27770, 352807, False, 1.5),
    "com-amazon-like": (334863, 1851744, True, 1.55),  # max degree ~550 like the SNAP graph
    "reddit-like": (232965, 114615892, True, 1.5),
    "products-like": (2449029, 123718280, True, 1.5),
    "pubmed-like": (19717, 88648, True, 1.5),
}


# ----------------------------------------------------------------------------- MatrixMarket via the C ABI

def read_mtx(path, cache_dir=None):
    """Returns dict(nrows, ncols, nnz, row, col, val) with numpy arrays (0-based,
    sorted by (row, col); symmetric files expanded, self-loops/duplicates dropped).
    With ``cache_dir`` the parsed result is kept there as a binary file and reused."""
    coo = Coo()
    if cache_dir is None:
        check(lib.gespmm_mtx_read(str(path).encode(), ctypes.byref(coo)), "gespmm_mtx_read(%s)" % path)
    else:
        check(lib.gespmm_mtx_read_cached(str(path).encode(), str(cache_dir).encode(), ctypes.byref(coo)),
              "gespmm_mtx_read_cached(%s)" % path)
    try:
        n = int(coo.nnz)
        if n:
            row = np.ctypeslib.as_array(coo.row, shape=(n,)).copy()
            col = np.ctypeslib.as_array(coo.col, shape=(n,)).copy()
            val = np.ctypeslib.as_array(coo.val, shape=(n,)).copy()
        else:
            row = np.zeros(0, np.int32)
            col = np.zeros(0, np.int32)
            val = np.zeros(0, np.float32)
        return {"nrows": int(coo.nrows), "ncols": int(coo.ncols), "nnz": n, "row": row, "col": col, "val": val}
    finally:
        lib.gespmm_mtx_free(ctypes.byref(coo))


def coo_to_csr(nrows, ncols, row, col, val=None):
    """Counting sort by row, input order kept inside a row. val=None -> all ones
    (what the reference driver does with file values, spmm_test.cu:574)."""
    row = np.ascontiguousarray(row, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    nnz = row.shape[0]
    if col.shape[0] != nnz:
        raise ValueError("row and col must have the same length")
    rowptr = np.empty(nrows + 1, dtype=np.int32)
    colind = np.empty(max(nnz, 1), dtype=np.int32)
    vals = np.empty(max(nnz, 1), dtype=np.float32)
    vin = None
    if val is not None:
        vin = np.ascontiguousarray(val, dtype=np.float32)
    rc = lib.gespmm_coo_to_csr(nrows, ncols, nnz, row.ctypes.data, col.ctypes.data,
                               vin.ctypes.data if vin is not None else None, rowptr.ctypes.data,
                               colind.ctypes.data, vals.ctypes.data)
    check(rc, "gespmm_coo_to_csr")
    return rowptr, colind[:nnz], vals[:nnz]


def load_mtx_as_csr(path, use_values=False):
    coo = read_mtx(path)
    rowptr, colind, vals = coo_to_csr(coo["nrows"], coo["ncols"], coo["row"], coo["col"],
                                      coo["val"] if use_values else None)
    return {"M": coo["nrows"], "K": coo["ncols"], "nnz": coo["nnz"], "rowptr": rowptr, "colind": colind,
            "val": vals}


def row_partition(rowptr, parts):
    """nnz-balanced contiguous row ranges: returns cut[parts+1] (numpy int64)."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    cut = np.empty(parts + 1, dtype=np.int64)
    check(lib.gespmm_row_partition(rowptr.ctypes.data, rowptr.shape[0] - 1, parts, cut.ctypes.data),
          "gespmm_row_partition")
    return cut


# ----------------------------------------------------------------------------- synthetic graphs

def _coprime_multiplier(M, seed):
    a = (2654435761 * (seed + 1)) % M
    a = max(a, 1)
    while math.gcd(a, M) != 1:
        a += 1
    return a


def _endpoints(n, M, gamma, gen, device):
    """n node ids with a power-law-ish popularity: floor(M * u^gamma), then scrambled
    so that popular nodes are spread over the id range."""
    u = torch.rand(n, generator=gen, device=device, dtype=torch.float64)
    idx = torch.clamp((u.pow(gamma) * M).to(torch.int64), max=M - 1)
    return idx


def _scramble(idx, M, seed):
    a = _coprime_multiplier(M, seed)
    b = (97 * (seed + 7)) % M
    return (idx * a + b) % M


def synthetic_csr(M, nnz, symmetric=True, gamma=1.5, seed=42, device="cpu", locality=0.0, band=2000, K=None,
                  col_offset=0):
    """Seeded random graph with EXACTLY ``nnz`` stored entries and no duplicates.

    symmetric: undirected edges (i != j) stored in both directions (nnz must be even);
               otherwise directed, self-loops allowed.
    gamma:     popularity skew of the endpoints (1 = uniform, larger = heavier tail).
    locality:  fraction of edges whose second endpoint is drawn within a Laplace(band)
               window of the first (in id space) instead of by popularity. 0 = none.
    K / col_offset: rectangular use (multi-GPU row shards): columns live in
               [col_offset, col_offset + M) of a K-column matrix.
    Returns (rowptr int32[M+1], colind int32[nnz]) torch tensors on ``device``;
    column indices are ascending inside every row.
    """
    device = torch.device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    if symmetric and nnz % 2:
        raise ValueError("symmetric graphs need an even nnz")
    target = nnz // 2 if symmetric else nnz
    if target > (M * (M - 1) // 2 if symmetric else M * M):
        raise ValueError("nnz exceeds the number of distinct entries")
    keys = torch.empty(0, dtype=torch.int64, device=device)
    want = target
    rounds = 0
    while keys.numel() < target:
        rounds += 1
        n = int(want * 1.15) + 1024
        u = _scramble(_endpoints(n, M, gamma, gen, device), M, seed)
        v = _scramble(_endpoints(n, M, gamma, gen, device), M, seed)
        if locality > 0.0:
            is_local = torch.rand(n, generator=gen, device=device) < locality
            # Laplace(band) offset = difference of two exponentials
            e1 = -torch.log1p(-torch.rand(n, generator=gen, device=device, dtype=torch.float64))
            e2 = -torch.log1p(-torch.rand(n, generator=gen, device=device, dtype=torch.float64))
            off = ((e1 - e2) * band).to(torch.int64)
            v = torch.where(is_local, (u + off) % M, v)
        if symmetric:
            lo = torch.minimum(u, v)
            hi = torch.maximum(u, v)
            k = lo * M + hi
            k = k[lo != hi]
        else:
            k = u * M + v
        keys = torch.unique(torch.cat([keys, k]))
        want = max(target - keys.numel(), 0)
        if rounds > 64:
            raise RuntimeError("synthetic_csr did not converge")
    if keys.numel() > target:
        sel = torch.randperm(keys.numel(), generator=gen, device=device)[:target]
        keys = keys[sel]
    r = keys // M
    c = keys % M
    if symmetric:
        r, c = torch.cat([r, c]), torch.cat([c, r])
    order = torch.argsort(r * M + c)
    r = r[order]
    c = c[order]
    counts = torch.bincount(r, minlength=M)
    rowptr = torch.zeros(M + 1, dtype=torch.int64, device=device)
    rowptr[1:] = torch.cumsum(counts, 0)
    if K is not None:
        c = c + int(col_offset)
    return rowptr.to(torch.int32), c.to(torch.int32)



def community_csr(M, nnz, n_comm, n_groups, intra_deg, group_share=0.6, size_skew=1.5, gamma=1.55, seed=42,
                  device="cpu", shuffle=True, return_levels=False):
    """Symmetric graph with EXACTLY ``nnz`` stored entries and PLANTED two-level community structure whose
    vertex ids are then shuffled by a seeded random permutation — so any locality has to be found by the
    consumer (row clustering), it is not inherited from the generator.

      level 1   ``n_comm`` communities with skewed sizes (size of community c ~ c^(1/size_skew - 1), largest a few
                hundred vertices at com-Amazon's size, smallest 2-3): every pair inside a community of s vertices is
                an edge with probability min(1, intra_deg / (s - 1)) — small communities are near-cliques, which is
                what gives co-purchase networks their high clustering coefficient;
      level 2   communities are dealt at random to ``n_groups`` groups of ~M / n_groups vertices; of the remaining
                edges a share ``group_share`` joins a popularity-drawn vertex to a uniformly drawn member of its
                own group;
      global    the rest joins two popularity-drawn vertices anywhere (power-law-ish tail, exponent ``gamma``,
                as in synthetic_csr).

    Returns (rowptr int32[M+1], colind int32[nnz], planted int64[M]) — ``planted[v]`` orders the vertices by
    (group, community), i.e. argsort(planted) is the order a perfect community detector would produce. With
    ``return_levels`` two more int64[M] follow: the community and the group of every vertex (after the shuffle).
    """
    device = torch.device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    if nnz % 2:
        raise ValueError("symmetric graphs need an even nnz")
    target = nnz // 2
    u = torch.rand(M, generator=gen, device=device, dtype=torch.float64)
    c1 = torch.clamp((u.pow(size_skew) * n_comm).to(torch.int64), max=n_comm - 1)
    c1, _ = torch.sort(c1)
    _, c1 = torch.unique(c1, return_inverse=True)  # compact labels, still ascending with the planted id
    nc1 = int(c1.max()) + 1
    size1 = torch.bincount(c1, minlength=nc1)
    off1 = torch.zeros(nc1 + 1, dtype=torch.int64, device=device)
    off1[1:] = torch.cumsum(size1, 0)
    ids = torch.arange(M, device=device)
    # level 1: every pair (i < j) of a community, kept with probability q(s)
    pos = ids - off1[c1]
    s_of = size1[c1]
    cnt = s_of - 1 - pos
    src = torch.repeat_interleave(ids, cnt)
    start = torch.cumsum(cnt, 0) - cnt
    dst = src + 1 + (torch.arange(src.numel(), device=device) - torch.repeat_interleave(start, cnt))
    q = torch.clamp(float(intra_deg) / torch.clamp(s_of[src] - 1, min=1).to(torch.float64), max=1.0)
    keep = torch.rand(src.numel(), generator=gen, device=device, dtype=torch.float64) < q
    keys = torch.unique(src[keep] * M + dst[keep])
    del src, dst, keep, q, start
    if keys.numel() > target:
        raise ValueError("intra-community edges alone exceed nnz/2: lower intra_deg")
    # level 2: communities dealt to groups at random
    perm_c = torch.randperm(nc1, generator=gen, device=device)
    cum = torch.cumsum(size1[perm_c], 0)
    g_of_c = torch.empty(nc1, dtype=torch.int64, device=device)
    g_of_c[perm_c] = torch.clamp((cum - 1) * n_groups // M, max=n_groups - 1)
    c2 = g_of_c[c1]
    members2 = torch.argsort(c2, stable=True)
    off2 = torch.zeros(n_groups + 1, dtype=torch.int64, device=device)
    off2[1:] = torch.cumsum(torch.bincount(c2, minlength=n_groups), 0)
    want = target - keys.numel()
    rounds = 0
    while want > 0:
        rounds += 1
        n = int(want * 1.2) + 1024
        a = _scramble(_endpoints(n, M, gamma, gen, device), M, seed)
        t = torch.rand(n, generator=gen, device=device)
        r = torch.rand(n, generator=gen, device=device, dtype=torch.float64)
        ga = c2[a]
        v_group = members2[off2[ga] + (r * (off2[ga + 1] - off2[ga]).to(torch.float64)).to(torch.int64)]
        v_glob = _scramble(torch.clamp((r.pow(gamma) * M).to(torch.int64), max=M - 1), M, seed)
        b = torch.where(t < group_share, v_group, v_glob)
        lo, hi = torch.minimum(a, b), torch.maximum(a, b)
        k = torch.unique((lo * M + hi)[lo != hi])
        k = k[~torch.isin(k, keys)]
        if k.numel() > want:
            k = k[torch.randperm(k.numel(), generator=gen, device=device)[:want]]
        keys = torch.cat([keys, k])
        want = target - keys.numel()
        if rounds > 64:
            raise RuntimeError("community_csr did not converge")
    r = keys // M
    c = keys % M
    planted = c2 * nc1 + c1
    comm_of, group_of = c1, c2
    if shuffle:
        P = torch.randperm(M, generator=gen, device=device)
        r, c = P[r], P[c]
        pl = torch.empty_like(planted)
        pl[P] = planted
        planted = pl
        comm_of, group_of = planted % nc1, planted // nc1
    r, c = torch.cat([r, c]), torch.cat([c, r])
    order = torch.argsort(r * M + c)
    r, c = r[order], c[order]
    rowptr = torch.zeros(M + 1, dtype=torch.int64, device=device)
    rowptr[1:] = torch.cumsum(torch.bincount(r, minlength=M), 0)
    if return_levels:
        return rowptr.to(torch.int32), c.to(torch.int32), planted, comm_of, group_of
    return rowptr.to(torch.int32), c.to(torch.int32), planted


# Structured stand-ins (same M and nnz as the structureless ones above). com-Amazon: SNAP lists 334 863 nodes,
# 925 872 edges, 75 149 ground-truth communities, average clustering coefficient 0.3967; ogbn-products (also an
# Amazon co-purchase graph): 2 449 029 nodes, 61 859 140 edges, average clustering coefficient 0.411 (OGB).
# (n_comm, n_groups, intra_deg, group_share) are set so that the generated graph reproduces the clustering
# coefficient (measured: scripts/reorder_study.py --stats); the group level is an assumption (product categories).
COMMUNITY_SPECS = {
    "com-amazon-sbm": ("com-amazon-like", 75149, 1024, 5.0, 0.6),
    "products-sbm": ("products-like", 51000, 2048, 34.0, 0.6),
    # the real reddit graph is made of subreddits; the structureless stand-in has none: 290 communities of ~800 rows, ~330 of a row's
    # 492 entries inside (profiles/r03/dense_community_audit.log)
    "reddit-sbm": ("reddit-like", 290, 16, 330.0, 0.6),
}

# Power-law graphs WITHOUT planted communities — the families the plan's clustering finds least in (ba-m6 runs at 0.13 of
# the roofline, holme-kim-m16 at 0.07). The hold-out audit (scripts/holdout_graphs.py) takes them from networkx; these are the same
# models generated on the device so that bench.py can carry them in every record without files: (vertices, edges per new vertex,
# probability of a triad-formation step).
PREFERENTIAL_SPECS = {
    "ba-m6": (500000, 6, 0.0),           # Barabasi-Albert: pure preferential attachment (hubs, clustering ~0)
    "holme-kim-m16": (400000, 16, 0.6),  # Holme-Kim: preferential attachment + triad formation (triangles, no communities)
}


def preferential_csr(n, m, p_triad=0.0, seed=42, device="cpu", batch=2000):
    """Growing-network model, batched: vertices arrive `batch` at a time and attach `m` edges each to the graph as it was when their
    batch began. An edge is a preferential-attachment step — the endpoint of a uniformly drawn earlier half-edge, i.e. a vertex with
    probability proportional to its degree (Barabasi-Albert) — or, after the first and with probability `p_triad`, a triad-formation
    step: a uniformly drawn neighbour of the previous target (Holme-Kim, networkx.powerlaw_cluster_graph's model). Returns a
    symmetric CSR without self loops or repeated entries, vertex ids shuffled by a seeded permutation."""
    device = torch.device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed) * 7919 + n + 31 * m)
    cap = (n + 1) * m + m + 1
    src = torch.empty(cap, dtype=torch.int64, device=device)
    dst = torch.empty(cap, dtype=torch.int64, device=device)
    E = m + 1  # seed graph: a ring over the first m + 1 vertices
    src[:E] = torch.arange(E, device=device)
    dst[:E] = (torch.arange(E, device=device) + 1) % E
    t = m + 1
    while t < n:
        b = max(1, min(batch, n - t, t // 32))  # (a batch never outweighs the graph it attaches to: early vertices arrive almost one by one)
        if p_triad > 0.0:  # adjacency of the graph so far (both directions), for the triad steps
            s2 = torch.cat([src[:E], dst[:E]])
            d2 = torch.cat([dst[:E], src[:E]])
            order = torch.argsort(s2)
            adj = d2[order]
            ptr = torch.zeros(t + 1, dtype=torch.int64, device=device)
            ptr[1:] = torch.cumsum(torch.bincount(s2, minlength=t), 0)
        nodes = torch.arange(t, t + b, device=device)
        prev = None
        for k in range(m):
            r = torch.randint(0, 2 * E, (b,), generator=gen, device=device)
            tgt = torch.where(r < E, src[r % E], dst[r % E])
            if prev is not None and p_triad > 0.0:
                lo = ptr[prev]
                deg = ptr[prev + 1] - lo
                pick = adj[lo + (torch.rand(b, generator=gen, device=device) * deg).long().clamp(max=deg - 1)]
                tgt = torch.where(torch.rand(b, generator=gen, device=device) < p_triad, pick, tgt)
            src[E + k * b:E + (k + 1) * b] = nodes
            dst[E + k * b:E + (k + 1) * b] = tgt
            prev = tgt
        E += m * b
        t += b
    gperm = torch.Generator(device="cpu")
    gperm.manual_seed(int(seed) + 1234)
    perm = torch.randperm(n, generator=gperm).to(device)
    u, v = perm[src[:E]], perm[dst[:E]]
    keep = u != v
    u, v = u[keep], v[keep]
    key = torch.unique(torch.cat([u * n + v, v * n + u]))  # symmetric, entries once, sorted by (row, column)
    rows, cols = key // n, key % n
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=n), 0)
    return rowptr.to(torch.int32), cols.to(torch.int32)


def synthetic_graph(name, seed=42, device="cpu", locality=0.0, band=2000, scale=1.0):
    """One of the named stand-ins. ``scale`` < 1 shrinks M and nnz
    proportionally (CPU-sized tests); scale == 1 reproduces the exact M / nnz."""
    if name in PREFERENTIAL_SPECS:
        n, m, p = PREFERENTIAL_SPECS[name]
        n = max(int(n * scale), 4 * m + 8)
        rowptr, colind = preferential_csr(n, m, p, seed, device)
        return {"name": name, "M": n, "K": n, "nnz": int(colind.numel()), "rowptr": rowptr, "colind": colind}
    if name in COMMUNITY_SPECS:
        base, n_comm, n_groups, intra_deg, group_share = COMMUNITY_SPECS[name]
        M, nnz, _, gamma = SPECS[base]
        if scale != 1.0:
            M = max(int(M * scale), 64)
            nnz = max(int(nnz * scale), 64)
            nnz -= nnz % 2
            n_comm = max(int(n_comm * scale), 4)
            n_groups = max(int(n_groups * scale), 2)
        rowptr, colind, planted, comm_of, group_of = community_csr(M, nnz, n_comm, n_groups, intra_deg, group_share, 1.5,
                                                                   gamma, seed, device, return_levels=True)
        return {"name": name, "M": M, "K": M, "nnz": int(colind.numel()), "rowptr": rowptr, "colind": colind,
                "truth": planted, "truth_community": comm_of, "truth_group": group_of}
    if name == "pubmed-selfloop-like":
        M, nnz, sym, gamma = SPECS["pubmed-like"]
    else:
        M, nnz, sym, gamma = SPECS[name]
    if scale != 1.0:
        M = max(int(M * scale), 8)
        nnz = max(int(nnz * scale), 8)
        if sym:
            nnz -= nnz % 2
    rowptr, colind = synthetic_csr(M, nnz, sym, gamma, seed, device, locality, band)
    if name == "pubmed-selfloop-like":  # gcn_custom.py:29-36 adds one self-loop per node
        rowptr, colind = add_self_loops(rowptr, colind)
    return {"name": name, "M": M, "K": M, "nnz": int(colind.numel()), "rowptr": rowptr, "colind": colind}


def add_self_loops(rowptr, colind):
    """A + I on a CSR pattern without existing self-loops (columns stay sorted)."""
    M = rowptr.numel() - 1
    dev = rowptr.device
    counts = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
    rows = torch.repeat_interleave(torch.arange(M, device=dev), counts)
    r = torch.cat([rows, torch.arange(M, device=dev)])
    c = torch.cat([colind.to(torch.int64), torch.arange(M, device=dev)])
    order = torch.argsort(r * M + c)
    r = r[order]
    c = c[order]
    newptr = torch.zeros(M + 1, dtype=torch.int64, device=dev)
    newptr[1:] = torch.cumsum(torch.bincount(r, minlength=M), 0)
    return newptr.to(torch.int32), c.to(torch.int32)


def relabel_by_order(rowptr, colind, order):
    """The square matrix with vertex order[i] renamed to i (rows AND columns), columns sorted inside every row: what a caller who keeps
    the graph in a community order hands over. ``order`` is a permutation of range(M) (e.g. argsort of the planted labels)."""
    M = rowptr.numel() - 1
    dev = rowptr.device
    order = order.to(dev).long()
    new_id = torch.empty(M, dtype=torch.int64, device=dev)
    new_id[order] = torch.arange(M, device=dev)
    deg = (rowptr[1:] - rowptr[:-1]).long()
    rows_old = torch.repeat_interleave(torch.arange(M, device=dev), deg)
    r, c = new_id[rows_old], new_id[colind.long()]
    idx = torch.argsort(r * M + c)
    r, c = r[idx], c[idx]
    rp = torch.zeros(M + 1, dtype=torch.int64, device=dev)
    rp[1:] = torch.cumsum(torch.bincount(r, minlength=M), 0)
    return rp.to(torch.int32), c.to(torch.int32)


def transpose_csr(rowptr, colind, K=None, val=None):
    """CSC arrays (colptr, rowind[, cscval]) of a CSR pattern, rows ascending inside a
    column — host-side helper for callers that build both orders once, like
    gcn_custom.py:39-46 does with scipy. Pure index manipulation with torch ops."""
    M = rowptr.numel() - 1
    K = M if K is None else K
    dev = rowptr.device
    counts = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
    rows = torch.repeat_interleave(torch.arange(M, device=dev), counts)
    key = colind.to(torch.int64) * M + rows
    order = torch.argsort(key)
    colptr = torch.zeros(K + 1, dtype=torch.int64, device=dev)
    colptr[1:] = torch.cumsum(torch.bincount(colind.to(torch.int64), minlength=K), 0)
    rowind = rows[order].to(torch.int32)
    if val is not None:
        return colptr.to(torch.int32), rowind, val[order]
    return colptr.to(torch.int32), rowind


def reference_B(K, N, seed=1, device="cpu"):
    """Dense operand with the reference driver's value set, float(r % 100 - 50) / 100
    (spmm_test.cu:586-594), from a seeded torch generator instead of libc rand()."""
    gen = torch.Generator(device="cpu")
    gen.manual_seed(int(seed))
    r = torch.randint(0, 100, (K, N), generator=gen, dtype=torch.int32)
    return ((r - 50).to(torch.float32) / 100).to(device)


# ----------------------------------------------------------------------------- RMAT (Graph500 Kronecker)

def _vertex_scramble(v, scale, seed):
    """Bijection on [0, 2^scale): odd multiplier, xor-shift, odd multiplier (all mod 2^scale).
    Spreads the RMAT hubs over the id range so contiguous row shards carry equal work."""
    mask = (1 << scale) - 1
    m1 = (0x9E3779B97F4A7C15 * (2 * seed + 1)) & mask | 1
    m2 = (0xBF58476D1CE4E5B9 * (2 * seed + 3)) & mask | 1
    sh = max(scale // 2, 1)
    v = (v * m1) & mask
    v = v ^ (v >> sh)
    v = (v * m2) & mask
    return v


def _rmat_chunk(scale, n, a, b, c, seed, chunk, device):
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed) * 1000003 + chunk)
    rows = torch.zeros(n, dtype=torch.int64, device=device)
    cols = torch.zeros(n, dtype=torch.int64, device=device)
    for _ in range(scale):
        u = torch.rand(n, generator=gen, device=device)
        rbit = (u >= a + b).to(torch.int64)
        cbit = (((u >= a) & (u < a + b)) | (u >= a + b + c)).to(torch.int64)
        rows = rows * 2 + rbit
        cols = cols * 2 + cbit
    return _vertex_scramble(rows, scale, seed), _vertex_scramble(cols, scale, seed)


def rmat_shard(scale, edge_factor=16, rank=0, world=1, seed=42, device="cpu", probs=(0.57, 0.19, 0.19, 0.05),
               chunk_edges=1 << 25, balanced=True):
    """One contiguous row range of an RMAT graph with M = 2^scale vertices and
    edge_factor*M directed edges (Graph500 parameters by default), as a CSR shard whose
    column indices address the full M columns.

    Every rank generates the same global edge stream chunk by chunk from a seeded
    counter (chunk index -> generator seed) and keeps its own rows, so the shards are
    consistent without any communication. With ``balanced`` a first pass over the
    stream histograms the row degrees and the row ranges are the nnz-balanced cuts
    (same rule as gespmm_row_partition); otherwise rows are split evenly. Duplicate
    edges are kept (a general CSR may repeat a column), so the global nnz is exactly
    edge_factor * 2^scale.
    Returns dict(M, K, nnz, rowptr, colind, row_begin, row_end, global_nnz)."""
    device = torch.device(device)
    M = 1 << scale
    total = edge_factor * M
    a, b, c, _ = probs
    nchunks = (total + chunk_edges - 1) // chunk_edges
    sizes = [min(chunk_edges, total - i * chunk_edges) for i in range(nchunks)]
    if balanced and world > 1:
        deg = torch.zeros(M, dtype=torch.int64, device=device)
        for ch, n in enumerate(sizes):
            rows, _ = _rmat_chunk(scale, n, a, b, c, seed, ch, device)
            deg += torch.bincount(rows, minlength=M)
        cum = torch.cumsum(deg, 0)  # cum[r] = rowptr[r+1]
        del deg
        targets = torch.tensor([(total * p) // world for p in range(1, world)], dtype=torch.int64, device=device)
        # first row r with rowptr[r] >= target  <=>  first r with cum[r-1] >= target
        inner = (torch.searchsorted(cum, targets, right=False) + 1).clamp(max=M).tolist()
        cuts = [0] + inner + [M]
        for i in range(1, len(cuts)):
            cuts[i] = max(cuts[i], cuts[i - 1])
        del cum
    else:
        cuts = [(M * p) // world for p in range(world + 1)]
    r0, r1 = cuts[rank], cuts[rank + 1]
    keep_r, keep_c = [], []
    for ch, n in enumerate(sizes):
        rows, cols = _rmat_chunk(scale, n, a, b, c, seed, ch, device)
        sel = (rows >= r0) & (rows < r1)
        keep_r.append(rows[sel] - r0)
        keep_c.append(cols[sel])
    r = torch.cat(keep_r)
    cc = torch.cat(keep_c)
    del keep_r, keep_c
    order = torch.argsort(r * M + cc)
    r = r[order]
    cc = cc[order].to(torch.int32)
    nloc = r1 - r0
    rowptr = torch.zeros(nloc + 1, dtype=torch.int64, device=device)
    if nloc > 0:
        rowptr[1:] = torch.cumsum(torch.bincount(r, minlength=nloc), 0)
    return {"M": nloc, "K": M, "nnz": int(cc.numel()), "rowptr": rowptr.to(torch.int32), "colind": cc,
            "row_begin": r0, "row_end": r1, "global_nnz": total, "cuts": cuts}


def write_mtx(path, rowptr, colind, K=None):
    """Write a CSR pattern as a MatrixMarket `pattern general` file (1-based), e.g. to
    feed a synthetic stand-in to the spmm_test driver the way run_test.sh feeds it the
    SNAP files."""
    rp = rowptr.cpu().numpy() if hasattr(rowptr, "cpu") else np.asarray(rowptr)
    ci = colind.cpu().numpy() if hasattr(colind, "cpu") else np.asarray(colind)
    M = rp.shape[0] - 1
    K = M if K is None else K
    rows = np.repeat(np.arange(M, dtype=np.int64), np.diff(rp.astype(np.int64)))
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate pattern general\n")
        f.write("%% synthetic stand-in written by gespmm_amd.graphs.write_mtx\n")
        f.write("%d %d %d\n" % (M, K, ci.shape[0]))
        np.savetxt(f, np.stack([rows + 1, ci.astype(np.int64) + 1], axis=1), fmt="%d %d")
