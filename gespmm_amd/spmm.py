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
    """The analysis stage for ONE sparse matrix at one feature width (``gespmm_plan_*`` of the C ABI — what the
    vendor libraries call preprocess; the reference has none). Creating a plan analyses the matrix once, on the
    device (``analysis="host"`` keeps the round-2 host form: same order, ~20x slower), and keeps what every later
    launch reuses: the long-row decision from the longest row it
    saw, the split points of the cache-blocked path (dense graphs), and — for sparse graphs whose B exceeds the
    L2s — a row-CLUSTERED copy of the matrix with an nnz-balanced task table, so rows that share neighbours run
    next to each other and find the shared B rows in L2 (and, at N = 128 / 256 where blocks of 96 / 64 clustered rows
    reuse their B rows, the tables of the staged-rows kernel: those rows are read from LDS; made for THIS width
    only). Only the processing order changes: the result has the same bits as the plain call.
    ``kernel``: "auto" | "stream" | "seg-stream" | "staged" | "records" | "staged-slabs" (dense clustered matrices, N = 128).

        plan = SpmmPlan(rowptr, colind, K, N, values=val)      # reorder="auto" | True | False
        out = csr_spmm(rowptr, colind, val, dense, plan=plan)

    The plan keeps references to ``rowptr`` / ``colind`` (and the values it last saw) and notices in-place edits
    through the tensors' version counters: new VALUES are re-permuted automatically, a changed PATTERN raises —
    make a new plan. One plan serves one stream at a time.

    ``expected_launches`` (0 = 200, the reference's protocols): with ``reorder="auto"`` the analysis is weighed against the
    products that will use it — a matrix whose estimated gain x launches does not pay for the estimated analysis time keeps
    its storage order and costs one validation pass (``describe()`` says so).
    """

    def __init__(self, rowptr, colind, K, N, variant=_lib.VARIANT_AUTO, values=None, reorder="auto", task_entries=0,
                 threads=0, flags=0, row_floor=0, kernel="auto", analysis="device", expected_launches=0):
        _need(rowptr, "rowptr", torch.int32, 1)
        _need(colind, "colind", torch.int32, 1)
        if values is not None:
            _need(values, "values", torch.float32, 1)
            if values.numel() != colind.numel():
                raise ValueError("values and colind must have the same length")
        dev = _same_device(rowptr, colind) if values is None else _same_device(rowptr, colind, values)
        self.shape = (rowptr.numel() - 1, int(K), int(N), colind.numel(), int(variant))
        self._rowptr, self._colind = rowptr, colind  # strong references: the plan may point at them
        self._pattern_version = (rowptr._version, colind._version)
        self._values = values
        self._values_version = values._version if values is not None else None
        self.device = dev
        mode = {"auto": _lib.PLAN_REORDER_AUTO, True: _lib.PLAN_REORDER, False: _lib.PLAN_NO_REORDER}[reorder]
        kern = {"auto": _lib.PLAN_KERNEL_AUTO, "stream": _lib.PLAN_KERNEL_STREAM, "seg-stream": _lib.PLAN_KERNEL_SEG_STREAM,
                "staged": _lib.PLAN_KERNEL_STAGED, "records": _lib.PLAN_KERNEL_RECORDS, "staged-slabs": _lib.PLAN_KERNEL_STAGED_SLABS}[kernel]
        where = {"device": _lib.PLAN_ANALYSIS_DEVICE, "host": _lib.PLAN_ANALYSIS_HOST}[analysis]
        opt = _lib.PlanOptions(mode, int(task_entries), int(row_floor), int(threads), int(flags), kern, where, int(expected_launches))
        self._handle = ctypes.c_void_p()
        M, K_, N_, nnz, var = self.shape
        with _on_device(dev):
            _lib.ensure_init(dev.index if dev.index is not None else torch.cuda.current_device(), _stream(dev), shape=(M, K_, nnz, N_),
                             expected_launches=expected_launches, forced=(reorder is True))
            rc = lib.gespmm_plan_create_v2(ctypes.byref(self._handle), _ptr(rowptr), _ptr(colind),
                                           _ptr(values) if values is not None else None, M, K_, nnz, N_, var,
                                           ctypes.byref(opt), ctypes.sizeof(opt), _stream(dev))
        check(rc, "gespmm_plan_create_v2")

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value and lib is not None:  # (module globals are gone at interpreter shutdown)
            lib.gespmm_plan_destroy(h)
            self._handle = ctypes.c_void_p()

    def tune(self, dense, out=None, reps=3):
        """Kernel choice by measurement (gespmm_plan_tune): the plan's candidate kernels run ``reps`` times each on ``dense`` and the
        fastest is kept; returns the product (same bits whichever wins). ``dense`` must have the plan's width."""
        _need(dense, "dense", torch.float32, 2)
        M, K, N, _, _ = self.shape
        if tuple(dense.shape) != (K, N):
            raise ValueError("tune() needs dense of shape (K, N) = (%d, %d)" % (K, N))
        if out is None:
            out = torch.empty((M, N), dtype=torch.float32, device=self.device)
        with _on_device(self.device):
            check(lib.gespmm_plan_tune(self._handle, _ptr(dense), _ptr(out), N, int(reps), _stream(self.device)), "gespmm_plan_tune")
        return out

    def describe(self):
        buf = ctypes.create_string_buffer(1400)
        n = lib.gespmm_plan_describe(self._handle, buf, 1400)
        if n < 0:
            check(int(n), "gespmm_plan_describe")
        return buf.value.decode()

    @property
    def clustered(self):
        return self.describe().startswith("order=clustered")

    def order(self):
        """perm[i] = row processed at position i (torch int32 on the CPU)."""
        perm = torch.empty(self.shape[0], dtype=torch.int32)
        rc = lib.gespmm_plan_get_order(self._handle, ctypes.c_void_p(perm.data_ptr()))
        if rc < 0:
            check(rc, "gespmm_plan_get_order")
        return perm

    def _sync_inputs(self, rowptr, colind, values, dense, variant):
        M, K, N, nnz, var = self.shape
        if (rowptr.numel() - 1, dense.shape[0], colind.numel(), int(variant)) != (M, K, nnz, var) or \
                rowptr.data_ptr() != self._rowptr.data_ptr() or colind.data_ptr() != self._colind.data_ptr():
            raise ValueError("SpmmPlan was made for a different matrix or variant")
        if (self._rowptr._version, self._colind._version) != self._pattern_version:
            raise ValueError("rowptr/colind were modified in place after the plan was made: create a new SpmmPlan")
        if values is None:
            if self._values is not None:
                with _on_device(self.device):
                    check(lib.gespmm_plan_set_values(self._handle, None, _stream(self.device)), "gespmm_plan_set_values")
                self._values, self._values_version = None, None
        elif self._values is None or values.data_ptr() != self._values.data_ptr() or \
                values._version != self._values_version:
            with _on_device(self.device):
                check(lib.gespmm_plan_set_values(self._handle, _ptr(values), _stream(self.device)), "gespmm_plan_set_values")
            self._values, self._values_version = values, values._version

    def run(self, values, dense, out=None, reduce_max=None):
        _need(dense, "dense", torch.float32, 2)
        M, K, _, _, _ = self.shape
        N = dense.shape[1]
        if _ext is not None and reduce_max is None and hasattr(_ext, "plan_spmm"):
            return _ext.plan_spmm(self._handle.value, dense, out, M)  # pybind11 path: ~5 us per call instead of ~12
        if out is None:
            out = torch.empty((M, N), dtype=torch.float32, device=self.device)
        with _on_device(self.device):
            if reduce_max is None:
                rc = lib.gespmm_plan_spmm_f32(self._handle, _ptr(dense), _ptr(out), N, _stream(self.device))
            else:
                rc = lib.gespmm_plan_spmm_max_f32(self._handle, _ptr(dense), _ptr(out), N, float(reduce_max),
                                                  _stream(self.device))
        check(rc, "gespmm_plan_spmm_f32")
        return out


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
        plan._sync_inputs(rowptr, colind, values, dense, variant)
        return plan.run(values, dense, out)
    cref = ctypes.byref(c) if c is not None else None
    # scratch for the cache-blocked / long-row paths from torch's allocator (see torch_binding.cpp)
    ws_bytes = lib.gespmm_csr_spmm_workspace_bytes(M, K, N, nnz, int(variant), cref)
    if ws_bytes < 0:
        check(int(ws_bytes), "gespmm_csr_spmm_workspace_bytes")
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes > 0 else None
    with _on_device(dev):
        rc = lib.gespmm_csr_spmm_f32_ws(_ptr(rowptr), _ptr(colind), _ptr(values) if values is not None else None,
                                        _ptr(dense), _ptr(out), M, K, N, nnz, int(variant), cref,
                                        _ptr(ws) if ws is not None else None, ws_bytes, _stream(dev))
    check(rc, "gespmm_csr_spmm_f32")
    return out


def csr_spmm(rowptr, colind, values, dense, variant=_lib.VARIANT_AUTO, cfg=None, out=None, plan=None):
    """C = A @ dense with A = CSR(rowptr, colind, values). Mirrors spmm.cpp:24-43."""
    if values is None:
        raise TypeError("csr_spmm needs edge values; use csr_spmm_no_edge_value for A == 1")
    if _ext is not None and cfg is None and out is None and plan is None:
        return _ext.csr_spmm(rowptr, colind, values, dense, int(variant))
    return _spmm(rowptr, colind, values, dense, variant, cfg, out, plan)


def csr_spmm_no_edge_value(rowptr, colind, dense, variant=_lib.VARIANT_AUTO, cfg=None, out=None, plan=None):
    """C = A @ dense with A == 1 on its pattern. Mirrors spmm.cpp:45-60."""
    if _ext is not None and cfg is None and out is None and plan is None:
        return _ext.csr_spmm_no_edge_value(rowptr, colind, dense, int(variant))
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


def baseline_copy(src, out=None):
    """dst[i] = src[i] through the library's streaming-copy yardstick (gespmm_baseline_copy_f32) — NOT a product path:
    bench.py times it beside the product to price `ceiling_frac` with the read + write rate of the box."""
    _need(src, "src", torch.float32, src.dim())
    if out is None:
        out = torch.empty_like(src)
    _need(out, "out", torch.float32, src.dim())
    if out.numel() != src.numel():
        raise ValueError("out must have as many elements as src")
    dev = _same_device(src, out)
    with _on_device(dev):
        rc = lib.gespmm_baseline_copy_f32(_ptr(src), _ptr(out), src.numel(), _stream(dev))
    check(rc, "gespmm_baseline_copy_f32")
    return out


def select_variant(M, nnz, N):
    """The variant VARIANT_AUTO resolves to for this shape (host-only)."""
    return lib.gespmm_select_variant(int(M), int(nnz), int(N))
