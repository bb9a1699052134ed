"""ctypes access to oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

May be imported by tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke();
never by the product package (gespmm_amd/). See the header of gespmm_oracle.c.
"""
import ctypes
import os
import subprocess
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_size_t, c_uint, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_PROBE = os.path.join(_HERE, "_ref", "mmio_probe")


def build(quiet=True):
    """(Re)build liboracle.so and, when /root/reference exists, oracle/_ref/."""
    subprocess.run(["make", "-C", _HERE] + (["-s"] if quiet else []), check=True)


def _load():
    if not os.path.exists(LIB_PATH):
        build()
    lib = ctypes.CDLL(LIB_PATH)
    ip, fp, dp = POINTER(c_int), POINTER(c_float), POINTER(c_double)
    for name in ("oracle_spmm_golden", "oracle_spmm_golden_omp", "oracle_spmm_fma"):
        getattr(lib, name).argtypes = [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        getattr(lib, name).restype = None
    lib.oracle_spmm_abs.argtypes = [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.oracle_spmm_max.argtypes = [c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p]
    lib.oracle_spmm_scatter.argtypes = [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.oracle_fill_B.argtypes = [c_uint, c_size_t, c_void_p]
    lib.oracle_coo_to_csr.argtypes = [c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                      c_void_p]
    lib.oracle_read_mtx.argtypes = [c_char_p, ip, ip, ip, POINTER(ip), POINTER(ip), POINTER(fp)]
    lib.oracle_read_mtx.restype = c_int
    lib.oracle_free.argtypes = [c_void_p]
    lib.oracle_sddmm.argtypes = [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p]
    lib.oracle_csr2csc.argtypes = [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.oracle_num_threads.restype = c_int
    return lib


lib = _load()


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data if a is not None else None


def spmm(rowptr, colind, val, B, mode="fma"):
    """mode: 'golden' (reference CPU loop, mul+add), 'fma' (device arithmetic),
    'omp' (golden body, all cores)."""
    rowptr, colind, B = _i32(rowptr), _i32(colind), _f32(B)
    val = _f32(val) if val is not None else None
    M, N = rowptr.shape[0] - 1, B.shape[1]
    C = np.empty((M, N), dtype=np.float32)
    fn = {"golden": lib.oracle_spmm_golden, "fma": lib.oracle_spmm_fma, "omp": lib.oracle_spmm_golden_omp}[mode]
    fn(M, N, _p(rowptr), _p(colind), _p(val), _p(B), _p(C))
    return C


def spmm_abs(rowptr, colind, val, B):
    rowptr, colind, B = _i32(rowptr), _i32(colind), _f32(B)
    val = _f32(val) if val is not None else None
    M, N = rowptr.shape[0] - 1, B.shape[1]
    S = np.empty((M, N), dtype=np.float64)
    lib.oracle_spmm_abs(M, N, _p(rowptr), _p(colind), _p(val), _p(B), _p(S))
    return S


def spmm_max(rowptr, colind, B, init=-10000.0):
    rowptr, colind, B = _i32(rowptr), _i32(colind), _f32(B)
    M, N = rowptr.shape[0] - 1, B.shape[1]
    C = np.empty((M, N), dtype=np.float32)
    lib.oracle_spmm_max(M, N, _p(rowptr), _p(colind), _p(B), float(init), _p(C))
    return C


def spmm_scatter(rowptr, colind, B):
    rowptr, colind, B = _i32(rowptr), _i32(colind), _f32(B)
    M, N = rowptr.shape[0] - 1, B.shape[1]
    C = np.empty((M, N), dtype=np.float32)
    lib.oracle_spmm_scatter(M, N, _p(rowptr), _p(colind), _p(B), _p(C))
    return C


def fill_B_rand(seed, K, N):
    """libc srand/rand exactly as spmm_test.cu:586-594 (image-specific sequence)."""
    B = np.empty((K, N), dtype=np.float32)
    lib.oracle_fill_B(int(seed), K * N, _p(B))
    return B


def hash_B(K, N, seed=1):
    """Platform-independent B with the reference's value set {-0.50 .. 0.49}:
    B.flat[i] = float(splitmix64(seed, i) % 100 - 50) / 100."""
    i = np.arange(K * N, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = i + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    r = (z >> np.uint64(33)) % np.uint64(100)
    return ((r.astype(np.int64) - 50).astype(np.float32) / np.float32(100)).reshape(K, N)


def hash_val(n, seed=7):
    """Edge values U[-0.5, 0.5) on a 1/4096 grid (SURVEY.md §8 d4 second run)."""
    i = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = i * np.uint64(0xD1342543DE82EF95) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(32))) * np.uint64(0xBF58476D1CE4E5B9)
        z = z ^ (z >> np.uint64(29))
    r = (z >> np.uint64(40)) % np.uint64(4096)
    return ((r.astype(np.int64) - 2048).astype(np.float32) / np.float32(4096))


def coo_to_csr(nrows, row, col, val=None):
    row, col = _i32(row), _i32(col)
    nnz = row.shape[0]
    v = _f32(val) if val is not None else None
    indptr = np.empty(nrows + 1, dtype=np.int32)
    indices = np.empty(max(nnz, 1), dtype=np.int32)
    data = np.empty(max(nnz, 1), dtype=np.float32)
    lib.oracle_coo_to_csr(nrows, nnz, _p(row), _p(col), _p(v), 1 if v is not None else 0, _p(indptr), _p(indices),
                          _p(data))
    return indptr, indices[:nnz], data[:nnz]


def read_mtx(path):
    nrows, ncols, nvals = c_int(), c_int(), c_int()
    r, c, v = POINTER(c_int)(), POINTER(c_int)(), POINTER(c_float)()
    rc = lib.oracle_read_mtx(str(path).encode(), ctypes.byref(nrows), ctypes.byref(ncols), ctypes.byref(nvals),
                             ctypes.byref(r), ctypes.byref(c), ctypes.byref(v))
    if rc != 0:
        return {"rc": rc}
    n = nvals.value
    out = {
        "rc": 0,
        "nrows": nrows.value,
        "ncols": ncols.value,
        "nnz": n,
        "row": np.ctypeslib.as_array(r, shape=(max(n, 1),))[:n].copy(),
        "col": np.ctypeslib.as_array(c, shape=(max(n, 1),))[:n].copy(),
        "val": np.ctypeslib.as_array(v, shape=(max(n, 1),))[:n].copy(),
    }
    lib.oracle_free(r)
    lib.oracle_free(c)
    lib.oracle_free(v)
    return out


def sddmm(rows, colind, D1, D2, csr=False):
    rows, colind, D1, D2 = _i32(rows), _i32(colind), _f32(D1), _f32(D2)
    nnz, N = colind.shape[0], D1.shape[1]
    M = D1.shape[0]
    out = np.empty(max(nnz, 1), dtype=np.float32)
    scale = np.empty(max(nnz, 1), dtype=np.float64)
    lib.oracle_sddmm(1 if csr else 0, M, nnz, N, _p(rows), _p(colind), _p(D1), _p(D2), _p(out), _p(scale))
    return out[:nnz], scale[:nnz]


def csr2csc(M, K, rowptr, colind, val=None):
    rowptr, colind = _i32(rowptr), _i32(colind)
    v = _f32(val) if val is not None else None
    nnz = colind.shape[0]
    colptr = np.empty(K + 1, dtype=np.int32)
    rowind = np.empty(max(nnz, 1), dtype=np.int32)
    cv = np.empty(max(nnz, 1), dtype=np.float32)
    lib.oracle_csr2csc(M, K, _p(rowptr), _p(colind), _p(v), _p(colptr), _p(rowind), _p(cv))
    return colptr, rowind[:nnz], (cv[:nnz] if v is not None else None)


def ref_mmio_probe(path):
    """Runs the reference's own mmio.hpp parser (oracle/_ref/mmio_probe) when built.
    Returns dict(banner_rc, size_rc, M, N, nz, typecode) or None if unavailable."""
    if not os.path.exists(REF_PROBE):
        return None
    out = subprocess.run([REF_PROBE, str(path)], capture_output=True, text=True).stdout.split()
    if not out or out[0] == "nofile":
        return {"nofile": True}
    return {"banner_rc": int(out[0]), "size_rc": int(out[1]), "M": int(out[2]), "N": int(out[3]),
            "nz": int(out[4]), "typecode": out[5]}


def num_threads():
    return lib.oracle_num_threads()
