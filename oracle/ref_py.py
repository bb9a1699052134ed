"""Access to oracle/_ref/ — the REFERENCE's own lines compiled from /root/reference by
oracle/make_ref.sh. TEST INFRASTRUCTURE ONLY (same rule as oracle_py: tests/, bench.py's
baseline legs and smoke(); never the product package).

  host    libref_host.so / ref_readmtx : readMtx<float> (util.hpp), COO->CSR, B init and the CPU
          golden loop of spmm_test.cu — what pins oracle/gespmm_oracle.c
  kernels libref_kernels.so            : the reference's CUDA kernels compiled by hipcc for
          gfx950 (needs a GPU to run): spmmWrapper + spmm_test0..4, the topo / valued kernels
          of pytorch-custom/spmm_kernel.cu with the dispatch of spmm_cuda[_no_edge_value]

`/root/reference` is needed only to BUILD these (here); the binaries travel to the GPU box.
"""
import ctypes
import os
import subprocess
import tempfile
from ctypes import c_char_p, c_int, c_uint, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(_HERE, "_ref")
HOST_LIB = os.path.join(REF_DIR, "libref_host.so")
KERNEL_LIB = os.path.join(REF_DIR, "libref_kernels.so")
READMTX = os.path.join(REF_DIR, "ref_readmtx")


def build():
    subprocess.run(["bash", os.path.join(_HERE, "make_ref.sh")], check=True)


def available():
    return os.path.exists(HOST_LIB) and os.path.exists(READMTX)


def kernels_available():
    return os.path.exists(KERNEL_LIB)


_host = None
_kern = None


def host():
    global _host
    if _host is None:
        if not available() and os.path.exists("/root/reference/spmm_test.cu"):
            build()
        lib = ctypes.CDLL(HOST_LIB)
        lib.ref_read_mtx.argtypes = [c_char_p, c_void_p, c_void_p, c_void_p]
        lib.ref_read_mtx.restype = c_int
        lib.ref_tuple_count.restype = c_int
        lib.ref_copy_tuples.argtypes = [c_void_p, c_void_p, c_void_p]
        lib.ref_coo_to_csr.argtypes = [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        lib.ref_fill_B.argtypes = [c_uint, c_int, c_int, c_void_p]
        lib.ref_golden.argtypes = [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        _host = lib
    return _host


def kernels():
    """The reference kernels (.so with gfx950 code objects; loading needs the HIP runtime)."""
    global _kern
    if _kern is None:
        lib = ctypes.CDLL(KERNEL_LIB)
        lib.ref_spmm_wrapper.argtypes = [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_int]
        lib.ref_spmm_wrapper.restype = c_int
        lib.ref_warmup.argtypes = [c_int]
        lib.ref_warmup.restype = c_int
        lib.ref_spmm_cuda_no_edge_value.argtypes = [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int]
        lib.ref_spmm_cuda_no_edge_value.restype = c_int
        lib.ref_spmm_cuda.argtypes = [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int]
        lib.ref_spmm_cuda.restype = c_int
        _kern = lib
    return _kern


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def read_mtx(path):
    """readMtx<float> run as a process (the reference exit()s on bad files and reads one
    element past its vectors on some inputs — keep that out of the test process).
    Returns dict(rc=<exit status>, nrows, ncols, nvals, row, col, val)."""
    with tempfile.TemporaryDirectory() as d:
        dump = os.path.join(d, "t.bin")
        p = subprocess.run([READMTX, str(path), dump], capture_output=True, text=True)
        if p.returncode != 0:
            return {"rc": p.returncode, "stdout": p.stdout}
        M, N, nv, nt = (int(x) for x in p.stdout.strip().splitlines()[-1].split())
        raw = np.fromfile(dump, dtype=np.int32)
        row, col = raw[:nt].copy(), raw[nt:2 * nt].copy()
        val = raw[2 * nt:3 * nt].view(np.float32).copy()
    return {"rc": 0, "nrows": M, "ncols": N, "nvals": nv, "tuples": nt, "row": row, "col": col, "val": val,
            "stdout": p.stdout}


def coo_to_csr(nrows, ncols, row, col):
    """spmm_test.cu:558-581 — returns (A_indptr, A_indices, A_data) with A_data == 1."""
    row, col = _i32(row), _i32(col)
    nnz = row.shape[0]
    indptr = np.empty(nrows + 1, dtype=np.int32)
    indices = np.empty(max(nnz, 1), dtype=np.int32)
    data = np.empty(max(nnz, 1), dtype=np.float32)
    host().ref_coo_to_csr(nrows, ncols, nnz, row.ctypes.data, col.ctypes.data, indptr.ctypes.data,
                          indices.ctypes.data, data.ctypes.data)
    return indptr, indices[:nnz], data[:nnz]


def fill_B(seed, max_ncols, A_ncols):
    """spmm_test.cu:592-594 after srand(seed)."""
    B = np.empty((A_ncols, max_ncols), dtype=np.float32)
    host().ref_fill_B(int(seed), max_ncols, A_ncols, B.ctypes.data)
    return B


def golden(indptr, indices, data, B):
    """spmm_test.cu:596-604."""
    indptr, indices, data, B = _i32(indptr), _i32(indices), _f32(data), _f32(B)
    M, N = indptr.shape[0] - 1, B.shape[1]
    out = np.empty((M, N), dtype=np.float32)
    host().ref_golden(M, N, indptr.ctypes.data, indices.ctypes.data, data.ctypes.data, B.ctypes.data,
                      out.ctypes.data)
    return out


# ---- device side: torch tensors on the GPU in, torch tensor out --------------------------------

def spmm_wrapper(method, tile_row, rowptr, colind, val, B, out=None, sync=True):
    """spmmWrapper(method, tile_row, ...) of spmm_test.cu:456 on device tensors."""
    import torch

    M, N = rowptr.numel() - 1, B.shape[1]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=B.device)
    rc = kernels().ref_spmm_wrapper(method, tile_row, M, N, rowptr.data_ptr(), colind.data_ptr(), val.data_ptr(),
                                    B.data_ptr(), out.data_ptr(), 1 if sync else 0)
    if rc != 0:
        raise RuntimeError("reference spmmWrapper failed: hip error %d" % rc)
    return out


def spmm_cuda(rowptr, colind, val, B, out=None, sync=True):
    """spmm_cuda / spmm_cuda_no_edge_value (val None) of pytorch-custom/spmm_kernel.cu."""
    import torch

    M, N = rowptr.numel() - 1, B.shape[1]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=B.device)
    if val is None:
        rc = kernels().ref_spmm_cuda_no_edge_value(M, N, rowptr.data_ptr(), colind.data_ptr(), B.data_ptr(),
                                                   out.data_ptr(), 1 if sync else 0)
    else:
        rc = kernels().ref_spmm_cuda(M, N, rowptr.data_ptr(), colind.data_ptr(), val.data_ptr(), B.data_ptr(),
                                     out.data_ptr(), 1 if sync else 0)
    if rc != 0:
        raise RuntimeError("reference spmm_cuda failed: hip error %d" % rc)
    return out
