"""How much of the planted structure of the headline stand-in does the plan's clustering find? Host only (no GPU): the host form of the
clustering (identical order to the device form) and the exact LRU model of the XCD L2s, window = B rows that 3 MiB hold at N = 128.
    python profiles/r04/experiments/clustering_vs_planted.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np

from gespmm_amd import _lib, graphs

lib = _lib.lib
g = graphs.synthetic_graph("com-amazon-sbm", seed=42, device="cpu")
M = g["M"]
rp, ci = g["rowptr"].numpy(), g["colind"].numpy()
grp, com = g["truth_group"].numpy(), g["truth_community"].numpy()


def hits(perm, window=6144):
    if perm is None:
        return lib.gespmm_simulate_l2_hits(rp.ctypes.data, ci.ctypes.data, M, M, None, 8, window)
    perm = np.ascontiguousarray(perm, dtype=np.int32)
    return lib.gespmm_simulate_l2_hits(rp.ctypes.data, ci.ctypes.data, M, M, perm.ctypes.data, 8, window)


perm = np.empty(M, np.int32)
lv = ctypes.c_int32()
cl = (ctypes.c_int32 * 16)()
lib.gespmm_cluster_rows(rp.ctypes.data, ci.ctypes.data, M, M, 0, perm.ctypes.data, ctypes.byref(lv), cl)
pos = np.empty(M, np.int64)
pos[perm] = np.arange(M)
rng = np.random.RandomState(0)
rows = np.repeat(np.arange(M), np.diff(rp))
print("entries inside their row's community %.3f, inside its group %.3f; %d communities, %d groups" %
      ((com[rows] == com[ci]).mean(), (grp[rows] == grp[ci]).mean(), len(np.unique(com)), len(np.unique(grp))))
print("clustering: %d levels, clusters %s" % (lv.value, list(cl)[:lv.value]))
print("modelled L2 hit rate (8 slices, 6144-row LRU):")
print("  storage order (ids shuffled)                         %.3f" % hits(None))
print("  communities contiguous, communities in random order  %.3f" % hits(np.argsort(rng.permutation(com.max() + 1)[com], kind="stable")))
print("  the plan's clustering                                %.3f" % hits(perm))
print("  planted order (group, community)                     %.3f" % hits(np.argsort(g["truth"].numpy(), kind="stable")))
print("  groups contiguous, random order inside a group       %.3f" % hits(np.lexsort((rng.rand(M), grp))))
print("  the plan's order, re-sorted by the TRUE group         %.3f" % hits(np.lexsort((pos, grp))))
pairs = len(np.unique(grp[rows].astype(np.int64) * M + ci))
for N in (32, 128, 512):
    lines = (4 * N + 127) // 128
    floor = 128 * lines * pairs + 4 * M * N + 4 * (M + 1) + 8 * len(ci)
    alg = 4 * (M + 1) + 8 * len(ci) + 8 * M * N
    print("N=%-3d traffic floor (every B row once per group that refers to it: %d pairs) %d B = %.2fx algorithmic; ceiling_frac %.3f" %
          (N, pairs, floor, floor / alg, alg / (floor / 5600.0) / 8000.0))
