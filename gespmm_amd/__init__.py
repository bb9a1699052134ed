"""gespmm_amd — MI355X-native GE-SpMM (CSR x dense fp32 SpMM) behind the reference's
PyTorch op surface.

Layout (only what the hot path needs):
  csrc/      HIP kernels (gfx950) + the C ABI of include/gespmm.h + host loader
  lib/       build output: libgespmm.so, spmm_test (git-ignored; built by __graft_entry__.build())
  _lib.py    ctypes binding of the C ABI — fails loudly if the library is missing
  spmm.py    mirror of the reference's pybind module `spmm`   (pytorch-custom/spmm.cpp:96-101)
  sddmm.py   mirror of the reference's pybind module `sddmm`  (pytorch-custom/sddmm.cpp:62-67)
  op.py      SPMMFunction / GCNConv                           (pytorch-custom/op.py)
  graphs.py  MatrixMarket loading via the C ABI + seeded synthetic stand-in graphs
  dist.py    1-D row partition + B exchange over torch.distributed (RCCL)

There is no CPU compute path anywhere in this package.
"""
from . import _lib  # noqa: F401  (loads libgespmm.so, raises if it is not built)
from . import spmm, sddmm, graphs  # noqa: F401
from .op import SPMMFunction, GCNConv  # noqa: F401

__all__ = ["spmm", "sddmm", "graphs", "SPMMFunction", "GCNConv"]
