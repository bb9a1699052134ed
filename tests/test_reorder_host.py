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


REC_WORDS, REC_ENTRIES, REC_DISTINCT, REC_ROWS = 96, 32, 16, 16  # csrc/spmm_kernels.h


def _records(lib, rowptr, colind, M, K, perm, target=0):
    recs = ctypes.POINTER(ctypes.c_int32)()
    src = ctypes.POINTER(ctypes.c_int32)()
    nrec = ctypes.c_int32(0)
    fn = lib.gespmm_debug_build_records
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32,
                   ctypes.POINTER(ctypes.POINTER(ctypes.c_int32)), ctypes.POINTER(ctypes.POINTER(ctypes.c_int32)),
                   ctypes.POINTER(ctypes.c_int32)]
    rc = fn(rowptr.ctypes.data, colind.ctypes.data, M, K, perm.ctypes.data, target, ctypes.byref(recs), ctypes.byref(src),
            ctypes.byref(nrec))
    assert rc == 0
    n = nrec.value
    r = np.ctypeslib.as_array(recs, shape=(n * REC_WORDS,)).copy().reshape(n, REC_WORDS)
    s = np.ctypeslib.as_array(src, shape=(n * REC_ENTRIES,)).copy().reshape(n, REC_ENTRIES)
    libc = ctypes.CDLL(None)
    libc.free.argtypes = [ctypes.c_void_p]
    libc.free(recs)
    libc.free(src)
    return r, s


def _interpret_records(recs, src, val, B, M):
    """What spmm_ldsrow.hip computes, restated with numpy in the kernel's order: one fused multiply-add per entry
    (float64 product + sum rounded once to float32 = fma for float operands), one chain per output element, chains
    of records carrying the accumulator."""
    N = B.shape[1]
    C = np.full((M, N), np.nan, dtype=np.float32)
    acc = None
    open_chain = False
    for i in range(len(recs)):
        rec = recs[i]
        nrows, nent, ndist, flags = int(rec[0]), int(rec[1]), int(rec[2]), int(rec[3])
        assert 0 <= nrows <= REC_ROWS and 0 <= nent <= REC_ENTRIES and 0 <= ndist <= REC_DISTINCT and 0 <= flags <= 3
        assert bool(flags & 1) == open_chain, "a record continues from the previous one iff that one continues into it"
        if flags:
            assert nrows == 1
        raw = rec.view(np.uint8)
        slots = raw[272:272 + REC_ENTRIES]
        rpb = raw[304:304 + REC_ROWS + 1]
        assert rpb[0] == 0 and rpb[nrows] == nent and np.all(np.diff(rpb[:nrows + 1].astype(int)) >= 0)
        dcols = rec[20:20 + REC_DISTINCT]
        assert len(set(dcols[:ndist].tolist())) == ndist, "distinct columns must be distinct"
        rows_lds = B[dcols[:ndist]]
        for r in range(nrows):
            if not (flags & 1):
                acc = np.zeros(N, dtype=np.float32)
            for k in range(rpb[r], rpb[r + 1]):
                assert slots[k] < ndist
                v = np.float32(val[src[i, k]]) if val is not None else np.float32(1.0)
                acc = (v.astype(np.float64) * rows_lds[slots[k]].astype(np.float64) + acc.astype(np.float64)).astype(np.float32)
            if not (flags & 2):
                crow = int(rec[4 + r])
                assert np.all(np.isnan(C[crow])), "every C row is written exactly once"
                C[crow] = acc
        open_chain = bool(flags & 2)
    assert not open_chain
    return C


def test_records_of_the_lds_rows_kernel_reproduce_the_oracle(pkg, oracle):
    """Host logic of the plan's second representation: records cut from a processing order, interpreted on the CPU
    in the kernel's arithmetic, must give the oracle's bits — incl. rows longer than a record, rows with more
    distinct columns than a record holds, duplicates inside a row, empty rows."""
    from gespmm_amd import _lib

    rng = np.random.RandomState(11)
    M, K = 700, 500
    deg = rng.geometric(0.15, size=M) - 1
    deg[5] = 300       # chain of records (entries)
    deg[6] = 25        # 17..32 entries with > 16 distinct columns: chain as well
    deg[7] = 32
    deg[100:140] = 0   # a run of empty rows
    rp = np.zeros(M + 1, dtype=np.int32)
    rp[1:] = np.cumsum(deg)
    ci = rng.randint(0, K, size=int(rp[-1])).astype(np.int32)
    ci[rp[6]:rp[7]] = np.arange(25) * 3
    ci[rp[7]:rp[8]] = rng.randint(0, 10, size=32)  # 32 entries over <= 10 distinct columns: fits ONE record
    val = oracle.hash_val(int(rp[-1]), seed=3)
    B = oracle.hash_B(K, 8, seed=4)
    for perm in (np.arange(M, dtype=np.int32), rng.permutation(M).astype(np.int32)):
        for target in (0, 8):
            recs, src = _records(_lib.lib, rp, ci, M, K, perm, target)
            got = _interpret_records(recs, src, val, B, M)
            ref = oracle.spmm(rp, ci, val, B, "fma")
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), target
            got_u = _interpret_records(recs, src, None, B, M)
            assert np.array_equal(got_u.view(np.uint32), oracle.spmm(rp, ci, None, B, "golden").view(np.uint32))
    recs, _ = _records(_lib.lib, rp, ci, M, K, np.arange(M, dtype=np.int32), 0)
    flags = recs[:, 3]
    assert (flags == 3).sum() >= 5 and (flags == 2).sum() >= 2 and (flags == 1).sum() == (flags == 2).sum()
    # row 7 (32 entries, <= 10 distinct columns) sits in ONE ordinary record of its own or with neighbours
    assert any(int(r[3]) == 0 and 7 in r[4:4 + int(r[0])].tolist() for r in recs)


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


def _outer_records(lib, rowptr, colind, M, K, perm, target=0):
    recs = ctypes.POINTER(ctypes.c_int32)()
    src = ctypes.POINTER(ctypes.c_int32)()
    nrec = ctypes.c_int32(0)
    fn = lib.gespmm_debug_build_outer_records
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32,
                   ctypes.POINTER(ctypes.POINTER(ctypes.c_int32)), ctypes.POINTER(ctypes.POINTER(ctypes.c_int32)),
                   ctypes.POINTER(ctypes.c_int32)]
    assert fn(rowptr.ctypes.data, colind.ctypes.data, M, K, perm.ctypes.data, target, ctypes.byref(recs), ctypes.byref(src),
              ctypes.byref(nrec)) == 0
    n = nrec.value
    r = np.ctypeslib.as_array(recs, shape=(n * 136,)).copy().reshape(n, 136)
    s = np.ctypeslib.as_array(src, shape=(n * 64,)).copy().reshape(n, 64)
    libc = ctypes.CDLL(None)
    libc.free.argtypes = [ctypes.c_void_p]
    libc.free(recs)
    libc.free(src)
    return r, s


def _interpret_outer_records(recs, src, val, B, M):
    """What spmm_outer.hip computes: per record the distinct columns in stored order, per column its (row, value) entries,
    acc[row] = fma(value, B[column], acc[row]); chains carry acc[0]."""
    N = B.shape[1]
    C = np.full((M, N), np.nan, dtype=np.float32)
    acc = np.zeros((8, N), dtype=np.float32)
    open_chain = False
    for i in range(len(recs)):
        rec = recs[i]
        nrows, nent, ndist, flags = int(rec[0]), int(rec[1]), int(rec[2]), int(rec[3])
        assert 1 <= nrows <= 8 and 0 <= nent <= 64 and 0 <= ndist <= 32 and 0 <= flags <= 3
        assert bool(flags & 1) == open_chain
        if flags:
            assert nrows == 1
        if not (flags & 1):
            acc[:] = 0
        raw = rec.view(np.uint8)
        erow = raw[432:432 + 64]
        cptr = raw[496:496 + 33]
        assert cptr[0] == 0 and cptr[ndist] == nent and np.all(np.diff(cptr[:ndist + 1].astype(int)) >= 0)
        dcol = rec[12:12 + 32]
        for j in range(ndist):
            b = B[dcol[j]].astype(np.float64)
            for e in range(cptr[j], cptr[j + 1]):
                r = int(erow[e])
                assert r < nrows
                v = np.float32(val[src[i, e]]) if val is not None else np.float32(1.0)
                acc[r] = (v.astype(np.float64) * b + acc[r].astype(np.float64)).astype(np.float32)
        if not (flags & 2):
            for r in range(nrows):
                crow = int(rec[4 + r])
                assert np.all(np.isnan(C[crow])), "every C row is written exactly once"
                C[crow] = acc[r]
        open_chain = bool(flags & 2)
    assert not open_chain
    return C


def test_records_of_the_task_outer_kernel_reproduce_the_oracle(pkg, oracle):
    """Column-major walk inside a task: rows with strictly ascending columns share a record (sorted union of their columns),
    everything else — unsorted rows, repeated columns, rows longer than a record — is a record or chain of its own in CSR
    order; either way every row's additions happen in its CSR order, so the interpretation equals the oracle's bits."""
    from gespmm_amd import _lib

    rng = np.random.RandomState(21)
    M, K = 600, 400
    deg = rng.geometric(0.2, size=M) - 1
    deg[3] = 200      # chain
    deg[4] = 40       # > 32 entries: chain even if sorted
    deg[50:70] = 0    # empty rows
    rp = np.zeros(M + 1, dtype=np.int32)
    rp[1:] = np.cumsum(deg)
    ci = np.empty(int(rp[-1]), dtype=np.int32)
    for r in range(M):
        d = deg[r]
        if r % 7 == 0:    # unsorted, possibly repeated columns (general .mtx rows)
            ci[rp[r]:rp[r + 1]] = rng.randint(0, K, size=d)
        else:             # sorted, distinct, drawn from a small neighbourhood so that rows share columns
            base = (r // 6) * 9 % (K - 40)
            ci[rp[r]:rp[r + 1]] = np.sort(rng.choice(np.arange(base, base + max(40, d + 1)) % K, size=d, replace=False)) if d else []
    val = oracle.hash_val(int(rp[-1]), seed=5)
    B = oracle.hash_B(K, 6, seed=6)
    for perm in (np.arange(M, dtype=np.int32), rng.permutation(M).astype(np.int32)):
        for target in (0, 12):
            recs, src = _outer_records(_lib.lib, rp, ci, M, K, perm, target)
            got = _interpret_outer_records(recs, src, val, B, M)
            assert np.array_equal(got.view(np.uint32), oracle.spmm(rp, ci, val, B, "fma").view(np.uint32)), target
            got_u = _interpret_outer_records(recs, src, None, B, M)
            assert np.array_equal(got_u.view(np.uint32), oracle.spmm(rp, ci, None, B, "golden").view(np.uint32))
    recs, _ = _outer_records(_lib.lib, rp, ci, M, K, np.arange(M, dtype=np.int32), 0)
    assert (recs[:, 0] > 1).sum() > 20, "rows are actually packed"
    assert (recs[:, 2] < recs[:, 1]).sum() > 10, "packed rows actually share columns"


def test_record_builders_reject_out_of_range_columns(pkg):
    """The host record builders index scratch arrays by column: an index outside [0, K) must be refused, not written."""
    from gespmm_amd import _lib

    lib = _lib.lib
    rp = np.array([0, 2, 3], dtype=np.int32)
    perm = np.array([1, 0], dtype=np.int32)
    for bad in (np.array([0, 5, 1], dtype=np.int32), np.array([0, -1, 1], dtype=np.int32)):
        for name in ("gespmm_debug_build_records", "gespmm_debug_build_outer_records"):
            fn = getattr(lib, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32,
                           ctypes.POINTER(ctypes.POINTER(ctypes.c_int32)), ctypes.POINTER(ctypes.POINTER(ctypes.c_int32)),
                           ctypes.POINTER(ctypes.c_int32)]
            recs = ctypes.POINTER(ctypes.c_int32)()
            src = ctypes.POINTER(ctypes.c_int32)()
            n = ctypes.c_int32(0)
            rc = fn(rp.ctypes.data, bad.ctypes.data, 2, 4, perm.ctypes.data, 0, ctypes.byref(recs), ctypes.byref(src), ctypes.byref(n))
            assert rc == -1, (name, rc)  # GESPMM_EINVAL
