"""Product host code (C ABI loader / COO->CSR / row partition) against the oracle."""
import glob
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN


def _all_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "*.mtx")) + glob.glob(os.path.join(GOLDEN, "mtx", "*.mtx")))


def test_loader_matches_oracle_on_every_fixture(pkg, oracle):
    from gespmm_amd import _lib, graphs

    n_ok = 0
    for path in _all_files():
        ref = oracle.read_mtx(path)
        if ref["rc"] != 0:
            with pytest.raises(_lib.GespmmError) as ei:
                graphs.read_mtx(path)
            assert ei.value.code == -5, path  # GESPMM_EFORMAT
            continue
        got = graphs.read_mtx(path)
        assert (got["nrows"], got["ncols"], got["nnz"]) == (ref["nrows"], ref["ncols"], ref["nnz"]), path
        assert np.array_equal(got["row"], ref["row"]), path
        assert np.array_equal(got["col"], ref["col"]), path
        assert np.array_equal(got["val"], ref["val"]), path
        n_ok += 1
    assert n_ok >= 10


def test_loader_handmade_expectations(pkg):
    from gespmm_amd import graphs

    with open(os.path.join(GOLDEN, "mtx_expected.json")) as f:
        expected = json.load(f)
    for name, exp in expected.items():
        if exp["rc"] == "format":
            continue
        got = graphs.read_mtx(os.path.join(GOLDEN, "mtx", name))
        assert got["row"].tolist() == exp["row"] and got["col"].tolist() == exp["col"], name
        assert np.array_equal(got["val"], np.array(exp["val"], dtype=np.float32)), name


def test_loader_errors_are_codes_not_exits(pkg, tmp_path):
    from gespmm_amd import _lib, graphs

    with pytest.raises(_lib.GespmmError) as ei:
        graphs.read_mtx(tmp_path / "missing.mtx")
    assert ei.value.code == -4  # GESPMM_EIO (reference: prints and exit(1), util.hpp:300-303)
    p = tmp_path / "nosize.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real general\n% only comments\n")
    with pytest.raises(_lib.GespmmError) as ei:
        graphs.read_mtx(p)
    assert ei.value.code == -5
    p = tmp_path / "array.mtx"
    p.write_text("%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n")
    with pytest.raises(_lib.GespmmError):
        graphs.read_mtx(p)
    p = tmp_path / "zero_based.mtx"
    p.write_text("%%MatrixMarket matrix coordinate pattern general\n2 2 1\n0 1\n")
    with pytest.raises(_lib.GespmmError):
        graphs.read_mtx(p)


def test_loader_large_random_file(pkg, oracle, tmp_path):
    """Ragged whitespace, many duplicates, symmetric expansion at a few 10^4 entries."""
    rng = np.random.RandomState(11)
    M, n = 700, 30000
    r = rng.randint(1, M + 1, n)
    c = rng.randint(1, M + 1, n)
    v = rng.randint(-9, 10, n)
    p = tmp_path / "rand_sym.mtx"
    with open(p, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate integer symmetric\n%%c\n%d %d %d\n" % (M, M, n))
        for i in range(n):
            f.write("%d\t%d   %d\n" % (r[i], c[i], v[i]) if i % 3 else " %d %d %d \n" % (r[i], c[i], v[i]))
    from gespmm_amd import graphs

    got, ref = graphs.read_mtx(p), oracle.read_mtx(p)
    assert got["nnz"] == ref["nnz"] and got["nnz"] > 0
    assert np.array_equal(got["row"], ref["row"]) and np.array_equal(got["col"], ref["col"])
    # duplicates with different values: both keep the first in (row, col, file order)
    assert np.array_equal(got["val"], ref["val"])


@pytest.mark.parametrize("kind", ["real general", "pattern symmetric", "real general, entries split over lines"])
def test_loader_multithreaded_parse(pkg, oracle, tmp_path, kind):
    """Files above 8 MB are parsed in pieces by several host threads and ordered by a counting sort +
    per-row sorts: same rows / cols / values as the oracle's serial restatement, in the same order
    (ties keep file order). A file whose entries are split over lines takes the serial path."""
    rng = np.random.RandomState(5)
    M, K, n = 50000, 40000, 700000
    r = rng.randint(1, M + 1, n)
    c = rng.randint(1, (M if "symmetric" in kind else K) + 1, n)
    c[::97] = c[1::97][: len(c[::97])]  # sprinkle repeats
    v = rng.randint(-999, 1000, n) / 8.0
    p = tmp_path / "big.mtx"
    field, sym = ("pattern", "symmetric") if "pattern" in kind else ("real", "general")
    with open(p, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate %s %s\n%% generated\n%d %d %d\n" % (field, sym, M, M if sym == "symmetric" else K, n))
        if field == "pattern":
            f.write("".join("%d %d\n" % (r[i], c[i]) for i in range(n)))
            f.write("".join("%d   %d\n" % (r[i], c[i]) for i in range(0, n, 2)))  # pad above 8 MB: extra lines past n are ignored
        elif "split" in kind:
            f.write("".join("%d\n%d %.3f\n" % (r[i], c[i], v[i]) for i in range(n)))
        else:
            f.write("".join("%d %d %.3f\n" % (r[i], c[i], v[i]) for i in range(n)))
    assert os.path.getsize(p) > (8 << 20)
    from gespmm_amd import graphs

    got, ref = graphs.read_mtx(p), oracle.read_mtx(p)
    assert ref["rc"] == 0 and got["nnz"] == ref["nnz"] and got["nnz"] > 0
    assert np.array_equal(got["row"], ref["row"]) and np.array_equal(got["col"], ref["col"])
    assert np.array_equal(got["val"], ref["val"])


def test_coo_to_csr_matches_oracle(pkg, oracle):
    from gespmm_amd import _lib, graphs

    rng = np.random.RandomState(5)
    nrows, ncols, nnz = 53, 61, 900
    row = rng.randint(0, nrows, nnz).astype(np.int32)
    col = rng.randint(0, ncols, nnz).astype(np.int32)
    val = rng.rand(nnz).astype(np.float32)
    for v in (None, val):
        a = graphs.coo_to_csr(nrows, ncols, row, col, v)
        b = oracle.coo_to_csr(nrows, row, col, v)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    ptr, ind, vv = graphs.coo_to_csr(4, 4, np.zeros(0, np.int32), np.zeros(0, np.int32))
    assert ptr.tolist() == [0, 0, 0, 0, 0] and ind.size == 0
    with pytest.raises(_lib.GespmmError):  # reference only prints "out of bound row" (spmm_test.cu:563)
        graphs.coo_to_csr(3, 3, np.array([3], np.int32), np.array([0], np.int32))
    with pytest.raises(_lib.GespmmError):
        graphs.coo_to_csr(3, 3, np.array([0], np.int32), np.array([3], np.int32))


def test_load_mtx_as_csr_forces_ones(pkg, bundled):
    from gespmm_amd import graphs

    for g in ("cora", "citeseer", "pubmed"):
        got = graphs.load_mtx_as_csr(os.path.join(GOLDEN, g + ".mtx"))
        assert np.array_equal(got["rowptr"], bundled[g]["rowptr"])
        assert np.array_equal(got["colind"], bundled[g]["colind"])
        assert np.all(got["val"] == 1.0)


def test_row_partition_properties(pkg, bundled):
    from gespmm_amd import graphs

    rowptr = bundled["pubmed"]["rowptr"]
    M, nnz = len(rowptr) - 1, int(rowptr[-1])
    for parts in (1, 2, 3, 8, 64):
        cut = graphs.row_partition(rowptr, parts)
        assert cut[0] == 0 and cut[-1] == M and np.all(np.diff(cut) >= 0)
        loads = rowptr[cut[1:]] - rowptr[cut[:-1]]
        assert loads.sum() == nnz
        maxdeg = int(np.diff(rowptr).max())
        assert loads.max() <= nnz / parts + maxdeg + 1, "balanced to within one row"
    skew = np.array([0, 0, 1000, 1000, 1001, 1002], dtype=np.int32)  # one giant row
    cut = graphs.row_partition(skew, 4)
    assert cut[0] == 0 and cut[-1] == 5 and np.all(np.diff(cut) >= 0)
    empty = np.zeros(6, dtype=np.int32)
    assert graphs.row_partition(empty, 3)[-1] == 5


def test_binary_cache_round_trip(pkg, tmp_path):
    import shutil
    import time

    from gespmm_amd import _lib, graphs

    src = tmp_path / "pubmed.mtx"
    shutil.copy(os.path.join(GOLDEN, "pubmed.mtx"), src)
    cache = tmp_path / "cache"
    cache.mkdir()
    t0 = time.perf_counter()
    a = graphs.read_mtx(src, cache_dir=cache)
    t_parse = time.perf_counter() - t0
    files = list(cache.iterdir())
    assert len(files) == 1 and files[0].name.startswith("pubmed.mtx.") and files[0].name.endswith(".gespmm-coo")
    t0 = time.perf_counter()
    b = graphs.read_mtx(src, cache_dir=cache)
    t_cached = time.perf_counter() - t0
    ref = graphs.read_mtx(src)
    for k in ("nrows", "ncols", "nnz"):
        assert a[k] == b[k] == ref[k]
    for k in ("row", "col", "val"):
        assert np.array_equal(a[k], ref[k]) and np.array_equal(b[k], ref[k])
    assert t_cached < t_parse
    # a changed file must not be served from the old cache
    with open(src, "a") as f:
        f.write("\n")
    os.utime(src, (time.time() + 5, time.time() + 5))
    graphs.read_mtx(src, cache_dir=cache)
    assert len(list(cache.iterdir())) == 2
    # a damaged cache file is ignored (fresh parse), an unwritable directory is not an error
    files[0].write_bytes(b"garbage")
    assert graphs.read_mtx(os.path.join(GOLDEN, "cora.mtx"), cache_dir=tmp_path / "does_not_exist")["nnz"] == 10556
    with pytest.raises(_lib.GespmmError):
        graphs.read_mtx(tmp_path / "missing.mtx", cache_dir=cache)


def test_loader_never_throws_across_the_c_boundary(pkg, tmp_path):
    """A size line that promises 10^9 entries in a 60-byte file must not reserve 24 GB (or std::terminate): the file
    size bounds the reservation and the missing entries are the reference's "not enough rows" case; an index outside
    the declared M x K is a malformed file, not a later out-of-bounds write."""
    from gespmm_amd import _lib, graphs

    p = tmp_path / "huge_promise.mtx"
    p.write_text("%%MatrixMarket matrix coordinate pattern general\n5 5 1000000000\n1 2\n3 4\n")
    coo = graphs.read_mtx(str(p))
    assert coo["nnz"] == 2 and list(coo["row"]) == [0, 2] and list(coo["col"]) == [1, 3]
    q = tmp_path / "out_of_range.mtx"
    q.write_text("%%MatrixMarket matrix coordinate pattern general\n3 3 2\n1 2\n4 1\n")
    with pytest.raises(_lib.GespmmError) as e:
        graphs.read_mtx(str(q))
    assert e.value.code == -5  # GESPMM_EFORMAT
