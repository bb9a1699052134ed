"""Mirror of the reference's pybind11 module ``sddmm`` (pytorch-custom/sddmm.cpp:21-67):

    coo_sddmm(rowind, colind, D1, D2) -> f32[nnz]     sddmm.cpp:21-40
    csr_sddmm(rowptr, colind, D1, D2) -> f32[nnz]     sddmm.cpp:42-60

out[e] = <D1[row(e), :], D2[col(e), :]> in pattern order (sddmm.cu:7-424). As in the
reference the feature width is D1.size(1) and, for the CSR form, M = D1.size(0).
"""
import torch

from ._ext import ext as _ext
from ._lib import check, lib
from .spmm import _need, _on_device, _ptr, _same_device, _stream


def _checked(idx0, name0, colind, D1, D2):
    _need(idx0, name0, torch.int32, 1)
    _need(colind, "colind", torch.int32, 1)
    _need(D1, "D1", torch.float32, 2)
    _need(D2, "D2", torch.float32, 2)
    if D1.shape[1] != D2.shape[1]:
        raise ValueError("D1 and D2 must have the same number of columns")
    return _same_device(D1, D2, idx0, colind)


def coo_sddmm(rowind, colind, D1, D2):
    if _ext is not None:
        return _ext.coo_sddmm(rowind, colind, D1, D2)
    dev = _checked(rowind, "rowind", colind, D1, D2)
    nnz = rowind.numel()
    if colind.numel() != nnz:
        raise ValueError("rowind and colind must have the same length")
    out = torch.empty((nnz,), dtype=torch.float32, device=dev)
    with _on_device(dev):
        rc = lib.gespmm_sddmm_coo_f32(_ptr(rowind), _ptr(colind), _ptr(D1), _ptr(D2), _ptr(out), nnz,
                                      D1.shape[1], _stream(dev))
    check(rc, "gespmm_sddmm_coo_f32")
    return out


def csr_sddmm(rowptr, colind, D1, D2, plan=None):
    """``plan``: a ``spmm.SpmmPlan`` of the same pattern — a clustered plan walks the edges in its own order (rows of D2
    shared by neighbouring rows come from L2) and returns the same bits in the caller's edge order."""
    if plan is not None:
        dev = _checked(rowptr, "rowptr", colind, D1, D2)
        if (rowptr.data_ptr(), colind.data_ptr()) != (plan._rowptr.data_ptr(), plan._colind.data_ptr()) or \
                (rowptr._version, colind._version) != plan._pattern_version:
            raise ValueError("the plan was made for a different (or since modified) pattern")
        if D1.shape[0] != rowptr.numel() - 1:
            raise ValueError("rowptr must have D1.size(0)+1 entries")
        if _ext is not None and hasattr(_ext, "plan_sddmm"):
            return _ext.plan_sddmm(plan._handle.value, D1, D2, colind.numel())
        out = torch.empty((colind.numel(),), dtype=torch.float32, device=dev)
        with _on_device(dev):
            rc = lib.gespmm_plan_sddmm_f32(plan._handle, _ptr(D1), _ptr(D2), _ptr(out), D1.shape[1], _stream(dev))
        check(rc, "gespmm_plan_sddmm_f32")
        return out
    if _ext is not None:
        return _ext.csr_sddmm(rowptr, colind, D1, D2)
    dev = _checked(rowptr, "rowptr", colind, D1, D2)
    M = D1.shape[0]
    if rowptr.numel() != M + 1:
        raise ValueError("rowptr must have D1.size(0)+1 entries")
    nnz = colind.numel()
    out = torch.empty((nnz,), dtype=torch.float32, device=dev)
    with _on_device(dev):
        rc = lib.gespmm_sddmm_csr_f32(_ptr(rowptr), _ptr(colind), _ptr(D1), _ptr(D2), _ptr(out), M, nnz,
                                      D1.shape[1], _stream(dev))
    check(rc, "gespmm_sddmm_csr_f32")
    return out
