#!/usr/bin/env python3
"""Host-side calibration of the plan's cost model (round 5): for every stand-in and hold-out graph, a CHEAP structural probe — the
share of sampled (row r, two of its columns c1, c2) triples with c2 in row c1: the local clustering coefficient on sampled wedges,
square matrices only — next to what the full analysis finds (modelled L2 hits in storage order -> clustered order, host form of the
clustering and the exact LRU model). The probe is what gespmm_plan_create can afford BEFORE deciding to cluster at all.
    python scripts/probe_calibration.py [graph ...]          (CPU only; minutes)"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from gespmm_amd import _lib, graphs  # noqa: E402

HOLD = os.environ.get("GESPMM_HOLDOUT_DIR", os.path.join(ROOT, "profiles", "r05", "holdout"))
lib = _lib.lib


def from_npz(path):
    z = np.load(path)
    n = int(z["n"])
    lo, hi = z["lo"].astype(np.int64), z["hi"].astype(np.int64)
    r, c = np.concatenate([lo, hi]), np.concatenate([hi, lo])
    order = np.argsort(r * n + c, kind="stable")
    r, c = r[order], c[order]
    rp = np.zeros(n + 1, dtype=np.int32)
    rp[1:] = np.cumsum(np.bincount(r, minlength=n))
    return n, rp, c.astype(np.int32)


def load(name):
    p = os.path.join(HOLD, name + ".npz")
    if os.path.exists(p):
        return from_npz(p)
    if name in ("cora", "citeseer", "pubmed"):
        g = graphs.load_mtx_as_csr(os.path.join(ROOT, "tests", "golden", name + ".mtx"))
        return g["M"], g["rowptr"], g["colind"]
    scale = 1.0
    if "@" in name:
        name, s = name.split("@")
        scale = float(s)
    g = graphs.synthetic_graph(name, seed=42, device="cpu", **({"scale": scale} if scale != 1.0 else {}))
    return g["M"], g["rowptr"].numpy().astype(np.int32), g["colind"].numpy().astype(np.int32)


def wedge_probe(M, rp, ci, samples=4096, pairs=4, seed=1):
    """P(c2 in row c1 | c1, c2 in row r), r sampled uniformly among rows with >= 2 entries; linear membership test (what one lane would do)."""
    rng = np.random.RandomState(seed)
    deg = np.diff(rp)
    cand = np.flatnonzero(deg >= 2)
    if cand.size == 0:
        return 0.0, 0
    rows = cand[rng.randint(0, cand.size, size=samples)]
    hit = tot = 0
    for r in rows:
        b, d = rp[r], deg[r]
        for _ in range(pairs):
            i, j = rng.randint(0, d), rng.randint(0, d - 1)
            if j >= i:
                j += 1
            c1, c2 = ci[b + i], ci[b + j]
            if c1 >= M:
                continue
            tot += 1
            hit += int(c2 in ci[rp[c1]:rp[c1 + 1]])
    return hit / max(tot, 1), tot


def analysis(M, rp, ci, window):
    perm = np.empty(M, dtype=np.int32)
    levels = ctypes.c_int32(0)
    clusters = (ctypes.c_int32 * 16)()
    t0 = time.time()
    rc = lib.gespmm_cluster_rows(rp.ctypes.data, ci.ctypes.data, M, M, 0, perm.ctypes.data, ctypes.byref(levels), clusters)
    assert rc == 0
    dt = time.time() - t0
    before = lib.gespmm_simulate_l2_hits(rp.ctypes.data, ci.ctypes.data, M, M, None, 8, window)
    after = lib.gespmm_simulate_l2_hits(rp.ctypes.data, ci.ctypes.data, M, M, perm.ctypes.data, 8, window)
    return before, after, dt


def main():
    names = sys.argv[1:] or ["com-amazon-sbm", "com-amazon-like", "geometric", "nws-k10", "lfr-mu0.1", "lfr-mu0.3", "holme-kim-m5", "ba-m6",
                             "pubmed", "products-sbm@0.1", "products-like@0.1"]
    print("%-20s %9s %10s %6s | %7s | %7s -> %7s  (N = 128: window 6144 rows)" % ("graph", "rows", "entries", "mean", "probe", "before", "after"))
    for name in names:
        M, rp, ci = load(name)
        rp, ci = np.ascontiguousarray(rp, dtype=np.int32), np.ascontiguousarray(ci, dtype=np.int32)
        cc, tot = wedge_probe(M, rp, ci)
        window = max(64, int(6144 * min(1.0, M / 300000.0))) if "@" in name or M < 100000 else 6144
        b, a, dt = analysis(M, rp, ci, window)
        print("%-20s %9d %10d %6.1f | %7.4f | %7.3f -> %7.3f  gain %.3f  (host clustering %.1f s, window %d)" %
              (name, M, ci.size, ci.size / M, cc, b, a, a - b, dt, window), flush=True)


if __name__ == "__main__":
    main()
