#!/usr/bin/env python3
"""Hold-out graphs for the plan heuristics (missing 3): generators this repository did NOT write and whose
parameters were not fitted against the plan's clustering — networkx 3.4 (LFR benchmark, Holme-Kim, Newman-Watts-Strogatz,
Barabasi-Albert, random geometric). Runs on the CPU (minutes), writes $GESPMM_HOLDOUT_DIR/<name>.npz (default profiles/r05/holdout) (edge list u < v,
vertex ids SHUFFLED by a seeded permutation so no locality is inherited from the generator's construction order).
    python scripts/holdout_graphs.py [name ...]
"""
import os
import sys
import time

import networkx as nx
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.environ.get("GESPMM_HOLDOUT_DIR", os.path.join(ROOT, "profiles", "r05", "holdout"))

CASES = {
    # LFR: power-law degrees (tau1) AND community sizes (tau2), mixing mu = share of a vertex's edges that leave its community
    "lfr-mu0.1": lambda: nx.LFR_benchmark_graph(300000, 2.5, 1.5, 0.1, average_degree=12, max_degree=300, min_community=20,
                                                max_community=1000, seed=11, max_iters=5000),
    "lfr-mu0.3": lambda: nx.LFR_benchmark_graph(300000, 2.5, 1.5, 0.3, average_degree=12, max_degree=300, min_community=20,
                                                max_community=1000, seed=12, max_iters=5000),
    "lfr-mu0.5": lambda: nx.LFR_benchmark_graph(300000, 2.5, 1.5, 0.5, average_degree=12, max_degree=300, min_community=20,
                                                max_community=1000, seed=13, max_iters=5000),
    "lfr-dense-mu0.3": lambda: nx.LFR_benchmark_graph(300000, 2.5, 1.5, 0.3, average_degree=40, max_degree=600, min_community=100,
                                                      max_community=2000, seed=14, max_iters=5000),
    # dense LFR (the reddit-shaped regime: mean degree in the hundreds): does the 0.65 rule for dense graphs hold off the stand-in?
    "lfr-verydense-mu0.2": lambda: nx.LFR_benchmark_graph(100000, 2.5, 1.5, 0.2, average_degree=250, max_degree=1500, min_community=500,
                                                          max_community=5000, seed=15, max_iters=5000),
    "lfr-verydense-mu0.5": lambda: nx.LFR_benchmark_graph(100000, 2.5, 1.5, 0.5, average_degree=250, max_degree=1500, min_community=500,
                                                          max_community=5000, seed=16, max_iters=5000),
    # Holme-Kim: preferential attachment + triad formation: triangles WITHOUT communities
    "holme-kim-m5": lambda: nx.powerlaw_cluster_graph(500000, 5, 0.6, seed=21),
    "holme-kim-m16": lambda: nx.powerlaw_cluster_graph(400000, 16, 0.6, seed=22),
    # small world: ring lattice (k nearest) + random shortcuts
    "nws-k10": lambda: nx.newman_watts_strogatz_graph(1000000, 10, 0.1, seed=31),
    # pure preferential attachment: hubs, no clustering
    "ba-m6": lambda: nx.barabasi_albert_graph(500000, 6, seed=41),
    # geometric: edges between points of the unit square closer than r (mean degree ~ n pi r^2 = 12)
    "geometric": lambda: nx.random_geometric_graph(600000, (12.0 / (600000 * np.pi)) ** 0.5, seed=51),
}


def main():
    os.makedirs(OUT, exist_ok=True)
    names = sys.argv[1:] or list(CASES)
    for name in names:
        t0 = time.time()
        try:
            g = CASES[name]()
        except Exception as ex:  # noqa: BLE001 - LFR may fail to converge for some parameters: say so and go on
            print("%-18s FAILED after %.0f s: %s: %s" % (name, time.time() - t0, type(ex).__name__, ex), flush=True)
            continue
        n = g.number_of_nodes()
        e = np.array([(u, v) for u, v in g.edges() if u != v], dtype=np.int64)
        perm = np.random.RandomState(1234).permutation(n)
        u, v = perm[e[:, 0]], perm[e[:, 1]]
        lo, hi = np.minimum(u, v).astype(np.int32), np.maximum(u, v).astype(np.int32)
        np.savez(os.path.join(OUT, name + ".npz"), n=np.int64(n), lo=lo, hi=hi)
        deg = np.bincount(np.concatenate([lo, hi]), minlength=n)
        print("%-18s n=%d entries=%d mean degree %.1f max %d  (%.0f s)" % (name, n, 2 * len(lo), deg.mean(), deg.max(), time.time() - t0),
              flush=True)


if __name__ == "__main__":
    main()
