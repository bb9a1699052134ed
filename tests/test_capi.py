"""The C ABI library: loads, exports everything include/gespmm.h declares, validates
its arguments on the host, and has no CPU compute path."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from helpers import ROOT


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "gespmm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gespmm_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(pkg):
    from gespmm_amd import _lib

    declared = _declared_functions()
    assert len(declared) >= 14
    assert sorted(_lib.EXPORTS) == declared, "ctypes binding and header disagree"
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    for name in declared:
        assert re.search(r"\bT %s\b" % name, nm), "%s not exported" % name
        getattr(_lib.lib, name)


def test_no_torch_types_in_header():
    text = open(os.path.join(ROOT, "include", "gespmm.h")).read()
    assert "torch" not in text.lower().replace("pytorch", "") and "at::" not in text


def test_version_and_error_strings(pkg):
    from gespmm_amd import _lib

    assert _lib.lib.gespmm_version().decode().startswith("gespmm ")
    assert _lib.lib.gespmm_error_string(0) == b"success"
    for code in (-1, -2, -3, -4, -5, -6):
        assert b"gespmm" in _lib.lib.gespmm_error_string(code)
    assert _lib.lib.gespmm_error_string(100)  # a hipError_t


def test_argument_validation_needs_no_gpu(pkg):
    from gespmm_amd import _lib

    lib = _lib.lib
    buf = (ctypes.c_int32 * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    odd = ctypes.c_void_p(p.value + 2)
    f = lib.gespmm_csr_spmm_f32
    assert f(p, p, None, p, p, -1, 4, 4, 0, -1, None) == -1          # negative M
    assert f(None, p, None, p, p, 4, 4, 4, 8, -1, None) == -1        # null rowptr
    assert f(p, p, None, p, None, 4, 4, 4, 8, -1, None) == -1        # null C
    assert f(p, None, None, p, p, 4, 4, 4, 8, -1, None) == -1        # null colind with nnz > 0
    assert f(p, p, None, p, p, 4, 4, 4, 8, 17, None) == -1           # unknown variant
    assert f(p, p, None, odd, p, 4, 4, 4, 8, -1, None) == -2         # misaligned B
    assert f(p, p, None, p, p, 1 << 33, 4, 4, 8, -1, None) == -3     # M beyond int32
    assert f(p, p, None, p, p, 0, 4, 4, 0, -1, None) == 0            # empty problem: no launch
    assert f(p, p, None, p, p, 4, 4, 0, 0, -1, None) == 0
    cfg = _lib.LaunchCfg(3, 0, 0, 0, 0, 0)
    assert lib.gespmm_csr_spmm_f32_cfg(p, p, None, p, p, 4, 4, 4, 8, -1, ctypes.byref(cfg), None) == -1
    cfg = _lib.LaunchCfg(0, 0, 24, 0, 0, 0)
    assert lib.gespmm_csr_spmm_f32_cfg(p, p, None, p, p, 4, 4, 4, 8, -1, ctypes.byref(cfg), None) == -1
    # max reducer exists for unweighted CRC variants only
    assert lib.gespmm_csr_spmm_max_f32(p, p, p, p, 4, 4, 4, 8, -1e4, 5, None) == -1
    assert lib.gespmm_sddmm_coo_f32(None, p, p, p, p, 8, 4, None) == -1
    assert lib.gespmm_sddmm_coo_f32(p, p, p, p, p, 0, 4, None) == 0
    assert lib.gespmm_sddmm_csr_f32(p, p, p, p, p, -1, 8, 4, None) == -1
    assert lib.gespmm_csr2csc_f32(None, p, None, p, p, None, 4, 4, 8, p, None) == -1
    assert lib.gespmm_csr2csc_f32(p, p, p, p, p, None, 4, 4, 8, p, None) == -1  # val in without val out
    assert lib.gespmm_row_partition(None, 4, 2, p) == -1


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful on a box without a GPU")
def test_no_cpu_fallback(pkg):
    """Without a HIP device the compute entry points fail loudly — there is no CPU path."""
    from gespmm_amd import _lib, spmm

    rowptr = np.array([0, 1, 2], dtype=np.int32)
    colind = np.array([0, 1], dtype=np.int32)
    B = np.ones((2, 4), dtype=np.float32)
    C = np.zeros((2, 4), dtype=np.float32)
    rc = _lib.lib.gespmm_csr_spmm_f32(rowptr.ctypes.data, colind.ctypes.data, None, B.ctypes.data, C.ctypes.data,
                                      2, 2, 4, 2, -1, None)
    assert rc > 0, "expected a hipError_t (no device)"
    assert np.all(C == 0), "nothing may be computed on the host"
    with pytest.raises(RuntimeError, match="no CPU path"):
        spmm.csr_spmm_no_edge_value(torch.from_numpy(rowptr), torch.from_numpy(colind), torch.from_numpy(B))
    from gespmm_amd import sddmm

    with pytest.raises(RuntimeError, match="no CPU path"):
        sddmm.coo_sddmm(torch.from_numpy(colind), torch.from_numpy(colind), torch.from_numpy(B),
                        torch.from_numpy(B))


def test_product_never_touches_the_oracle():
    """The product package and its C sources may not reference oracle/ in any way."""
    pkg_dir = os.path.join(ROOT, "gespmm_amd")
    for base, _, files in os.walk(pkg_dir):
        if os.sep + "lib" in base:
            continue
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                text = open(os.path.join(base, fn), errors="ignore").read().lower()
                assert "oracle" not in text, os.path.join(base, fn)
    ldd = subprocess.run(["ldd", os.path.join(pkg_dir, "lib", "libgespmm.so")], capture_output=True, text=True).stdout
    assert "oracle" not in ldd


def test_select_variant(pkg):
    from gespmm_amd import _lib, spmm

    assert spmm.select_variant(1000, 5000, 128) == _lib.VARIANT_CRC_CWM4
    assert spmm.select_variant(1000, 5000, 512) == _lib.VARIANT_CRC_CWM4
    assert spmm.select_variant(1000, 5000, 256) == _lib.VARIANT_CRC_CWM4
    assert spmm.select_variant(1000, 5000, 130) == _lib.VARIANT_CRC_CWM2
    assert spmm.select_variant(1000, 5000, 129) == _lib.VARIANT_CRC
    assert spmm.select_variant(1000, 5000, 64) == _lib.VARIANT_CRC
    assert spmm.select_variant(1000, 5000, 32) == _lib.VARIANT_CRC
    assert spmm.select_variant(1000, 5000, 6) == _lib.VARIANT_CRC
    assert spmm.select_variant(1000, 5000, 41) == _lib.VARIANT_CRC
    assert spmm.select_variant(1000, 5000, 3) == _lib.VARIANT_CRC
    for n in (1, 2, 3, 16, 41, 128, 500, 512):
        assert 0 <= spmm.select_variant(10, 10, n) <= 4, "auto never picks the tolerance-only variant"


def test_torch_extension_module(pkg):
    """The pybind11 extension (csrc/torch_binding.cpp) mirrors the reference's `spmm` and
    `sddmm` modules; it validates like the ctypes layer and has no CPU path either."""
    from gespmm_amd import _ext

    if _ext.ext is None:
        pytest.skip("extension not built (run __graft_entry__.build())")
    for name in ("csr_spmm", "csr_spmm_no_edge_value", "csr2csc", "coo_sddmm", "csr_sddmm", "csr_spmm_max"):
        assert callable(getattr(_ext.ext, name))
    rp = torch.tensor([0, 1, 2], dtype=torch.int32)
    ci = torch.tensor([0, 1], dtype=torch.int32)
    B = torch.ones(2, 4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        _ext.ext.csr_spmm_no_edge_value(rp, ci, B)
    with pytest.raises(RuntimeError, match="no CPU path"):
        _ext.ext.coo_sddmm(ci, ci, B, B)
    ldd = subprocess.run(["ldd", _ext.EXT_PATH], capture_output=True, text=True).stdout
    assert "libgespmm.so" in ldd and "oracle" not in ldd


def test_workspace_query_is_host_only(pkg):
    """gespmm_csr_spmm_workspace_bytes: pure host arithmetic on the shape — zero for problems that take
    a streaming kernel, split points for dense graphs, partial rows when the long-row pass is on."""
    import ctypes as C

    from gespmm_amd import _lib

    q = _lib.lib.gespmm_csr_spmm_workspace_bytes
    assert q(334863, 334863, 128, 1851744, -1, None) == 0          # com-Amazon-shaped: streaming kernel
    assert q(19717, 19717, 128, -1, -1, None) == 0                 # nnz unknown: nothing that needs scratch
    # reddit-shaped: dense -> (nslab + 1) * M int32 split points, 6 MB slabs of 512-byte rows
    M, nnz = 232965, 114615892
    nslab = -(-M // ((6 << 20) // 512))
    assert q(M, M, 128, nnz, -1, None) == (nslab + 1) * M * 4
    assert q(M, M, 128, nnz, 5, None) == 0                         # parallel-reduction variant: none
    # RMAT-shaped (mean degree 16, >= 2^23 entries): long-row pass, bounded by nnz/2048 chunks + nnz/2048 rows
    M, nnz, N = 1 << 22, 1 << 26, 128
    b = q(M, M, N, nnz, -1, None)
    chunks = nnz // 2048 + nnz // 2048 + 1
    assert chunks * N * 4 <= b <= chunks * (N * 4 + 8) + (nnz // 2048 + 1) * 16 + 1024
    # STRICT_ORDER switches the pass off
    c = _lib.LaunchCfg()
    c.flags = _lib.FLAG_STRICT_ORDER
    assert q(M, M, N, nnz, -1, C.byref(c)) == 0
    assert q(-1, 4, 4, 4, -1, None) < 0 and q(4, 4, 4, 4, 99, None) < 0


def test_describe_launch_pins_the_selection_rules(pkg):
    """gespmm_describe_launch is host-only: the kernel family and geometry AUTO picks for the
    BASELINE.json shapes are pinned here, so a change of the selection rules is a deliberate one
    (measurements behind them: profiles/r01/rows_per_wave_sweep.log, slab_size_sweep_v2.log, ...)."""
    import ctypes as C

    from gespmm_amd import _lib

    def describe(M, K, N, nnz, variant=-1, flags=0):
        buf = C.create_string_buffer(256)
        cfg = _lib.LaunchCfg(0, 0, 0, 0, 0, flags)
        n = _lib.lib.gespmm_describe_launch(M, K, N, nnz, variant, C.byref(cfg), buf, 256)
        assert n > 0
        return buf.value.decode()

    amazon = (334863, 334863, None, 1851744)
    assert describe(amazon[0], amazon[1], 128, amazon[3]) == \
        "variant=3 kernel=batch-stream V=4 S=1 W=32 rows_per_wave=4 idx32"
    assert describe(amazon[0], amazon[1], 32, amazon[3]) == \
        "variant=1 kernel=batch-stream V=1 S=1 W=32 rows_per_wave=16 idx32"
    reddit = (232965, 232965, None, 114615892)
    for N in (128, 256, 512):  # 512-byte column tiles bound to XCDs, 6 MB slabs
        assert describe(reddit[0], reddit[1], N, reddit[3]) == \
            "variant=3 kernel=slab-blocked V=4 S=1 W=32 slab_rows=12288 slabs=19 idx32"
    assert "kernel=batch-stream" in describe(reddit[0], reddit[1], 128, reddit[3], flags=_lib.FLAG_NO_SLAB_BLOCKED)
    # blocking needs >= ~20 entries of a row per slab: degree 150 over 17 slabs streams instead
    assert "kernel=batch-stream" in describe(200000, 200000, 128, 30000000)
    assert "kernel=batch-stream" in describe(20000, 20000, 256, 3000000)
    assert describe(1 << 22, 1 << 22, 128, 1 << 26).endswith("rows_per_wave=2 idx32 long_rows>2048 chunk=2048")
    assert "long_rows" not in describe(1 << 22, 1 << 22, 128, 1 << 26, flags=_lib.FLAG_STRICT_ORDER)
    assert "long_rows" in describe(1 << 16, 1 << 16, 128, 1 << 20)          # 2^20 entries at mean degree 16: on
    assert "long_rows" not in describe(1 << 16, 1 << 16, 128, (1 << 20) - 1)  # below 2^20: off
    assert "long_rows" not in describe(amazon[0], amazon[1], 128, amazon[3])  # mean degree 5.5: off below 2^23
    assert describe(1 << 26, 1 << 26, 256, 1 << 30).startswith("variant=3 kernel=batch-stream V=4 S=1 W=64")
    assert "idx64" in describe(1 << 26, 1 << 26, 256, 1 << 30)
    assert "kernel=segmented-stream" in describe(2048, 2048, 128, 10000)      # short rows, B resident in L2
    assert "c_stores" not in describe(300000, 2048, 128, 1500000)              # system-scope C stores are opt-in ...
    assert describe(300000, 2048, 128, 1500000, flags=0x8000).endswith("c_stores=sc1")  # ... GESPMM_FLAG_SC1_STORE
    assert "kernel=segmented-stream" in describe(1 << 20, 1 << 20, 128, 3 << 20)   # road-network-like: mean degree 3
    assert "kernel=batch-stream" in describe(1 << 20, 1 << 20, 128, 4 << 20)       # mean degree 4: batch kernel
    assert "kernel=batch-stream" in describe(1 << 15, 1 << 20, 128, 3 << 15)       # small matrix: batch kernel
    assert describe(reddit[0], reddit[1], 4, reddit[3], variant=5) == "variant=5 kernel=parallel-reduction W=64 idx32"
    assert "kernel=parallel-reduction" in describe(reddit[0], reddit[1], 8, reddit[3], flags=_lib.FLAG_ALLOW_REASSOCIATION)
    assert "kernel=parallel-reduction" not in describe(reddit[0], reddit[1], 8, reddit[3])
    assert describe(amazon[0], amazon[1], 260, amazon[3]).startswith("variant=4 kernel=batch-stream V=4 S=2 W=64")  # one 512-col tile
    assert describe(27770, 27770, 260, 352807).startswith("variant=3 ")         # small graph: two tiles
    assert describe(100, 100, 41, 1000).startswith("variant=1 ")               # odd N: one column per lane
    assert describe(100, 100, 130, 1000).startswith("variant=2 ") and " V=2 " in describe(100, 100, 130, 1000)
    buf = C.create_string_buffer(8)
    assert _lib.lib.gespmm_describe_launch(10, 10, 8, 20, -1, None, buf, 8) == 7  # truncated, NUL-terminated
    assert _lib.lib.gespmm_describe_launch(10, 10, 8, 20, 99, None, buf, 8) < 0
