"""Host side of the plan: row clustering (reorder.cpp) and the L2 model, through the C ABI with HOST pointers.
No device is needed: the clustering is pure host logic."""
import ctypes

import numpy as np
import pytest


def _cluster(lib, rowptr, colind, M, K, threads=0):
    perm = np.empty(M, dtype=np.int32)
    levels = ctypes.c_int32(0)
    clusters = (ctypes.c_int32 * 16)()
    rc = lib.gespmm_cluster_rows(rowptr.ctypes.data, colind.ctypes.data if colind.size else None, M, K, threads,
                                 perm.ctypes.data, ctypes.byref(levels), clusters)
    assert rc == 0
    return perm, levels.value, list(clusters)[:levels.value]


def _hits(lib, rowptr, colind, M, K, perm, window=512):
    return lib.gespmm_simulate_l2_hits(rowptr.ctypes.data, colind.ctypes.data, M, K,
                                       perm.ctypes.data if perm is not None else None, 8, window)


@pytest.fixture(scope="module")
def sbm(pkg):
    from gespmm_amd import graphs

    g = graphs.synthetic_graph("com-amazon-sbm", seed=42, device="cpu", scale=0.1)
    return g["M"], g["rowptr"].numpy().copy(), g["colind"].numpy().copy(), g["truth"].numpy()


def test_generator_contract(pkg, sbm):
    M, rp, ci, truth = sbm
    rows = np.repeat(np.arange(M), np.diff(rp))
    key = rows.astype(np.int64) * M + ci
    assert np.all(np.diff(key) > 0), "sorted, no duplicates"
    assert not np.any(rows == ci), "no self loops"
    assert np.array_equal(np.sort(ci.astype(np.int64) * M + rows), key), "symmetric"
    # ids are shuffled: the planted order is not the storage order
    assert np.mean(np.abs(np.diff(truth.astype(np.float64)))) > 0.1 * truth.max()


def test_clustering_is_a_permutation_and_deterministic(pkg, sbm):
    from gespmm_amd import _lib

    M, rp, ci, _ = sbm
    p1, levels, clusters = _cluster(_lib.lib, rp, ci, M, M, threads=1)
    assert np.array_equal(np.sort(p1), np.arange(M, dtype=np.int32))
    assert levels >= 2 and all(clusters[i] >= clusters[i + 1] for i in range(levels - 1))
    p4, _, _ = _cluster(_lib.lib, rp, ci, M, M, threads=4)
    p0, _, _ = _cluster(_lib.lib, rp, ci, M, M, threads=0)
    assert np.array_equal(p1, p4) and np.array_equal(p1, p0), "result must not depend on the thread count"


def test_clustering_finds_the_planted_communities(pkg, sbm):
    """Shuffled ids: storage order has ~no reuse; the clustered order must recover most of what the planted order
    offers under the L2 model (window scaled with the graph: 512 B rows per slice at 1/10 size)."""
    from gespmm_amd import _lib

    M, rp, ci, truth = sbm
    perm, _, _ = _cluster(_lib.lib, rp, ci, M, M)
    natural = _hits(_lib.lib, rp, ci, M, M, None)
    planted = _hits(_lib.lib, rp, ci, M, M, np.argsort(truth, kind="stable").astype(np.int32))
    found = _hits(_lib.lib, rp, ci, M, M, perm)
    assert natural < 0.05 and planted > 0.6
    assert found > 0.85 * planted, (natural, found, planted)


def test_l2_model_against_a_python_lru(pkg):
    from collections import OrderedDict

    from gespmm_amd import _lib

    rng = np.random.RandomState(3)
    M, K = 400, 300
    deg = rng.randint(0, 9, size=M)
    rp = np.zeros(M + 1, dtype=np.int32)
    rp[1:] = np.cumsum(deg)
    ci = rng.randint(0, K, size=int(rp[-1])).astype(np.int32)
    perm = rng.permutation(M).astype(np.int32)
    for order in (None, perm):
        got = _lib.lib.gespmm_simulate_l2_hits(rp.ctypes.data, ci.ctypes.data, M, K,
                                               order.ctypes.data if order is not None else None, 1, 37)
        od, hits = OrderedDict(), 0
        for r in (order if order is not None else range(M)):
            for c in ci[rp[r]:rp[r + 1]]:
                if c in od:
                    od.move_to_end(c)
                    hits += 1
                else:
                    od[c] = 1
                    if len(od) > 37:
                        od.popitem(last=False)
        assert abs(got - hits / rp[-1]) < 1e-12


def test_degenerate_inputs(pkg):
    from gespmm_amd import _lib

    # empty matrix, all-empty rows, one row, rectangular with out-of-range-free columns
    rp = np.zeros(1, dtype=np.int32)
    perm = np.empty(0, dtype=np.int32)
    assert _lib.lib.gespmm_cluster_rows(rp.ctypes.data, None, 0, 5, 0, perm.ctypes.data, None, None) == 0
    rp = np.zeros(11, dtype=np.int32)
    p, levels, _ = _cluster(_lib.lib, rp, np.zeros(0, np.int32), 10, 7)
    assert np.array_equal(np.sort(p), np.arange(10))
    rp = np.array([0, 3], dtype=np.int32)
    p, _, _ = _cluster(_lib.lib, rp, np.array([2, 0, 2], np.int32), 1, 3)
    assert list(p) == [0]
    rng = np.random.RandomState(1)
    M, K = 50, 2000
    deg = rng.randint(0, 6, size=M)
    rp = np.zeros(M + 1, dtype=np.int32)
    rp[1:] = np.cumsum(deg)
    ci = rng.randint(0, K, size=int(rp[-1])).astype(np.int32)
    p, _, _ = _cluster(_lib.lib, rp, ci, M, K)
    assert np.array_equal(np.sort(p), np.arange(M))
    assert _lib.lib.gespmm_cluster_rows(None, None, 5, 5, 0, p.ctypes.data, None, None) == -1


def test_clustering_on_the_bundled_real_graphs(pkg, bundled):
    """The only REAL graphs available offline (the reference's cora / citeseer / pubmed citation networks): at an L2
    window scaled to their size the clustered order multiplies the modelled reuse of the storage order — real graphs
    behave like the planted-community stand-in, not like the structureless one."""
    from gespmm_amd import _lib

    for name, floor in (("cora", 0.40), ("citeseer", 0.40), ("pubmed", 0.28)):
        g = bundled[name]
        perm, _, _ = _cluster(_lib.lib, g["rowptr"], g["colind"], g["M"], g["K"])
        before = _hits(_lib.lib, g["rowptr"], g["colind"], g["M"], g["K"], None, window=64)
        after = _hits(_lib.lib, g["rowptr"], g["colind"], g["M"], g["K"], perm, window=64)
        assert after >= floor and after >= 3 * before, (name, before, after)
