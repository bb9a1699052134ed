"""Mirror of the reference's pybind11 module ``spmm`` (pytorch-custom/spmm.cpp:96-101):

    csr_spmm(rowptr, colind, values, dense)        -> f32[M, N]   spmm.cpp:24-43
    csr_spmm_no_edge_value(rowptr, colind, dense)  -> f32[M, N]   spmm.cpp:45-60
    csr2csc(rowptr, colind, colptr, rowind, csr_data) -> f32[nnz] spmm.cpp:70-93

Same names, argument order and meaning. Where the reference only ``assert``s its
inputs (compiled out under NDEBUG) these functions raise. Outputs are allocated with
``torch.empty`` on ``dense.device`` like spmm_kernel.cu:183,434; kernels run on the
CURRENT torch stream (the reference uses the legacy default stream). Calls go through
the pybind11 extension `_gespmm_torch` (csrc/torch_binding.cpp) when it is built and
through the ctypes binding of the same C ABI otherwise or when tuning knobs are used;
both validate identically and neither has a CPU path. Extra keyword
arguments (``variant``, ``cfg``) expose the C ABI's tuning knobs and default to the
library's choice.
"""
import ctypes

import torch

from . import _lib
from ._ext import ext as _ext
from ._lib import LaunchCfg, check, lib


def _need(t, name, dtype, ndim):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if t.device.type != "cuda":
        raise RuntimeError("%s must be a HIP (cuda) device tensor; gespmm_amd has no CPU path" % name)
    if t.dtype != dtype:
        raise TypeError("%s must have dtype %s, got %s" % (name, dtype, t.dtype))
    if t.dim() != ndim:
        raise ValueError("%s must be %d-dimensional" % (name, ndim))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)


def _same_device(*ts):
    dev = ts[0].device
    for t in ts[1:]:
        if t.device != dev:
            raise RuntimeError("all tensors must live on the same device")
    return dev


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class _on_device:
    """`with torch.cuda.device(dev)` only when dev is not already current (the context
    manager costs ~3 us per op call; the common case needs nothing)."""

    __slots__ = ("ctx",)

    def __init__(self, dev):
        self.ctx = None if dev.index is None or dev.index == torch.cuda.current_device() else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False


def _make_cfg(cfg):
    if cfg is None:
        return None
    if isinstance(cfg, LaunchCfg):
        return cfg
    return LaunchCfg(int(cfg.get("vec", 0)), int(cfg.get("strips", 0)), int(cfg.get("group", 0)),
                     int(cfg.get("rows_per_wave", 0)), int(cfg.get("slab_rows", 0)), int(cfg.get("flags", 0)))


class SpmmPlan:
    """Scratch kept across calls for ONE sparse matrix at one feature width — the "analysis" stage of
    the vendor libraries. For dense graphs (cache-blocked path) it holds the per-row split points, which
    are computed on the first call and reused afterwards (0.2 ms per call on a reddit-sized graph). Creating
    a plan also looks at the longest row (one synchronisation) and switches the long-row pass on or off
    accordingly — rows that pass re-associates are within the 1e-4 tolerance, not bit-exact. The caller vouches that ``rowptr`` /
    ``colind`` do not change while the plan is in use and uses a plan on one stream at a time.

        plan = SpmmPlan(rowptr, colind, K, N)
        out = csr_spmm(rowptr, colind, values, dense, plan=plan)
    """

    def __init__(self, rowptr, colind, K, N, variant=_lib.VARIANT_AUTO):
        _need(rowptr, "rowptr", torch.int32, 1)
        _need(colind, "colind", torch.int32, 1)
        self.shape = (rowptr.numel() - 1, int(K), int(N), colind.numel(), int(variant))
        self.graph = (rowptr.data_ptr(), colind.data_ptr())
        # Analysis the plain entry points cannot afford (it synchronises): the longest row decides whether the
        # long-row pass is worth its launches — the library's own rule has to guess from nnz and the mean degree.
        M, nnz = self.shape[0], self.shape[3]
        self.flags = 0
        if M > 0 and nnz > 0:
            max_deg = int((rowptr[1:] - rowptr[:-1]).max().item())
            threshold = max(2048, 32 * ((nnz + M - 1) // M))
            self.flags = _lib.FLAG_SPLIT_LONG_ROWS if max_deg > threshold else _lib.FLAG_STRICT_ORDER
        self._cfg = LaunchCfg(0, 0, 0, 0, 0, self.flags)
        nbytes = lib.gespmm_csr_spmm_workspace_bytes(self.shape[0], self.shape[1], self.shape[2], self.shape[3],
                                                     int(variant), ctypes.byref(self._cfg))
        if nbytes < 0:
            check(int(nbytes), "gespmm_csr_spmm_workspace_bytes")
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=rowptr.device) if nbytes > 0 else None
        self.ready = False  # True once a call has written the split points

    def _flags_for(self, rowptr, colind, dense, variant):
        M, K, N, nnz, var = self.shape
        if (rowptr.numel() - 1, dense.shape[0], dense.shape[1], colind.numel(), int(variant)) != (M, K, N, nnz, var) or \
                (rowptr.data_ptr(), colind.data_ptr()) != self.graph:
            raise ValueError("SpmmPlan was made for a different matrix, width or variant")
        return self.flags | (_lib.FLAG_REUSE_SPLIT if (self.ready and self.workspace is not None) else 0)


def _spmm(rowptr, colind, values, dense, variant, cfg, out, plan=None):
    _need(rowptr, "rowptr", torch.int32, 1)
    _need(colind, "colind", torch.int32, 1)
    _need(dense, "dense", torch.float32, 2)
    if values is not None:
        _need(values, "values", torch.float32, 1)
        if values.numel() != colind.numel():
            raise ValueError("values and colind must have the same length")
        dev = _same_device(dense, rowptr, colind, values)
    else:
        dev = _same_device(dense, rowptr, colind)
    if rowptr.numel() < 1:
        raise ValueError("rowptr must have M+1 >= 1 entries")
    M = rowptr.numel() - 1
    K, N = dense.shape
    nnz = colind.numel()
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=dev)
    else:
        _need(out, "out", torch.float32, 2)
        if tuple(out.shape) != (M, N) or out.device != dev:
            raise ValueError("out must be f32[M, N] on the same device")
    c = _make_cfg(cfg)
    if plan is not None:
        if c is not None:
            raise ValueError("a plan fixes the launch configuration: pass either cfg or plan")
        c = LaunchCfg(0, 0, 0, 0, 0, plan._flags_for(rowptr, colind, dense, variant))
    cref = ctypes.byref(c) if c is not None else None
    # scratch for the cache-blocked / long-row paths from torch's allocator (see torch_binding.cpp)
    if plan is not None:
        ws = plan.workspace
        ws_bytes = ws.numel() if ws is not None else 0
    else:
        ws_bytes = lib.gespmm_csr_spmm_workspace_bytes(M, K, N, nnz, int(variant), cref)
        if ws_bytes < 0:
            check(int(ws_bytes), "gespmm_csr_spmm_workspace_bytes")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes > 0 else None
    with _on_device(dev):
        rc = lib.gespmm_csr_spmm_f32_ws(_ptr(rowptr), _ptr(colind), _ptr(values) if values is not None else None,
                                        _ptr(dense), _ptr(out), M, K, N, nnz, int(variant), cref,
                                        _ptr(ws) if ws is not None else None, ws_bytes, _stream(dev))
    check(rc, "gespmm_csr_spmm_f32")
    if plan is not None:
        plan.ready = True
    return out


def csr_spmm(rowptr, colind, values, dense, variant=_lib.VARIANT_AUTO, cfg=None, out=None, plan=None):
    """C = A @ dense with A = CSR(rowptr, colind, values). Mirrors spmm.cpp:24-43."""
    if values is None:
        raise TypeError("csr_spmm needs edge values; use csr_spmm_no_edge_value for A == 1")
    if _ext is not None and cfg is None and out is None:
        if plan is None:
            return _ext.csr_spmm(rowptr, colind, values, dense, int(variant))
        res = _ext.csr_spmm(rowptr, colind, values, dense, int(variant), plan.workspace,
                            plan._flags_for(rowptr, colind, dense, variant))
        plan.ready = True
        return res
    return _spmm(rowptr, colind, values, dense, variant, cfg, out, plan)


def csr_spmm_no_edge_value(rowptr, colind, dense, variant=_lib.VARIANT_AUTO, cfg=None, out=None, plan=None):
    """C = A @ dense with A == 1 on its pattern. Mirrors spmm.cpp:45-60."""
    if _ext is not None and cfg is None and out is None:
        if plan is None:
            return _ext.csr_spmm_no_edge_value(rowptr, colind, dense, int(variant))
        res = _ext.csr_spmm_no_edge_value(rowptr, colind, dense, int(variant), plan.workspace,
                                          plan._flags_for(rowptr, colind, dense, variant))
        plan.ready = True
        return res
    return _spmm(rowptr, colind, None, dense, variant, cfg, out, plan)


def csr_spmm_max(rowptr, colind, dense, empty_value=-10000.0, variant=_lib.VARIANT_AUTO):
    """C[r, :] = max over neighbours of dense[col, :] (DGL max reducer,
    binary_reduce_max.cu:182-207; rows without neighbours give ``empty_value``, the
    reference's hard-coded -10000)."""
    if _ext is not None:
        return _ext.csr_spmm_max(rowptr, colind, dense, float(empty_value), int(variant))
    _need(rowptr, "rowptr", torch.int32, 1)
    _need(colind, "colind", torch.int32, 1)
    _need(dense, "dense", torch.float32, 2)
    dev = _same_device(dense, rowptr, colind)
    M = rowptr.numel() - 1
    K, N = dense.shape
    out = torch.empty((M, N), dtype=torch.float32, device=dev)
    with _on_device(dev):
        rc = lib.gespmm_csr_spmm_max_f32(_ptr(rowptr), _ptr(colind), _ptr(dense), _ptr(out), M, K, N,
                                         colind.numel(), float(empty_value), int(variant), _stream(dev))
    check(rc, "gespmm_csr_spmm_max_f32")
    return out


def csr2csc(rowptr, colind, colptr, rowind, csr_data):
    """Fill ``colptr`` / ``rowind`` in place with the CSC form of CSR(rowptr, colind)
    and return the values in CSC order. Mirrors spmm.cpp:70-93 (whose CUDA
    implementation is unusable as shipped: spmm_kernel.cu:386 uses an uninitialised
    cuSPARSE handle). The number of columns is ``colptr.numel() - 1``."""
    if _ext is not None:
        return _ext.csr2csc(rowptr, colind, colptr, rowind, csr_data)
    _need(rowptr, "rowptr", torch.int32, 1)
    _need(colind, "colind", torch.int32, 1)
    _need(colptr, "colptr", torch.int32, 1)
    _need(rowind, "rowind", torch.int32, 1)
    _need(csr_data, "csr_data", torch.float32, 1)
    dev = _same_device(rowptr, colind, colptr, rowind, csr_data)
    M = rowptr.numel() - 1
    K = colptr.numel() - 1
    nnz = colind.numel()
    if rowind.numel() != nnz or csr_data.numel() != nnz:
        raise ValueError("rowind and csr_data must have nnz entries")
    out = torch.empty((nnz,), dtype=torch.float32, device=dev)
    with _on_device(dev):
        ws_bytes = lib.gespmm_csr2csc_workspace_bytes(M, K, nnz)
        if ws_bytes < 0:
            check(int(ws_bytes), "gespmm_csr2csc_workspace_bytes")
        ws = torch.empty((max(int(ws_bytes), 1),), dtype=torch.uint8, device=dev)
        rc = lib.gespmm_csr2csc_f32(_ptr(rowptr), _ptr(colind), _ptr(csr_data), _ptr(colptr), _ptr(rowind),
                                    _ptr(out), M, K, nnz, _ptr(ws), _stream(dev))
    check(rc, "gespmm_csr2csc_f32")
    return out


def select_variant(M, nnz, N):
    """The variant VARIANT_AUTO resolves to for this shape (host-only)."""
    return lib.gespmm_select_variant(int(M), int(nnz), int(N))
