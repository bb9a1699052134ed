"""The HIP path against THE REFERENCE'S OWN KERNELS running on the same MI355X.

oracle/_ref/libref_kernels.so is the reference's CUDA source compiled by hipcc for gfx950
from /root/reference by oracle/make_ref.sh (spmm_test.cu:62-492 — spmm_test0..4 + spmmWrapper;
pytorch-custom/spmm_kernel.cu:23-173, 210-379 — the topo / valued kernels of the torch op,
with the launch shapes of spmm_cuda[_no_edge_value] restated). It is the checker, never the
product. What these tests pin:
  * the product's results are IDENTICAL (bit for bit) to the reference's device results on the
    same inputs — every method 0..4 of the driver, both dispatchers of the torch op;
  * the oracle's `fma` restatement of the device arithmetic (what every other -m gpu test
    compares with) is what the reference's kernels actually compute, and the committed
    `valued_fma` vectors of tests/golden/spmm_checksums.json are the reference kernels' output.
"""
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, bits, edge_case_csr

import ref_py

# oracle/_ref/*.so are built where /root/reference exists and travel to the GPU box with the snapshot. A box without them
# SKIPS this file (visible with -rs: scripts/gpu_check.sh passes it) — unless GESPMM_REQUIRE_REF=1 says they must be
# there (smoke() sets it when it found them at build time: tests/golden/ref_built.stamp), in which case absence is a FAILURE.
_HAVE_REF = ref_py.kernels_available()
_REQUIRE_REF = os.environ.get("GESPMM_REQUIRE_REF") == "1" or os.path.exists(os.path.join(os.path.dirname(__file__), os.pardir,
                                                                                             "oracle", "_ref", "built.stamp"))
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not _HAVE_REF and not _REQUIRE_REF,
                                 reason="REFERENCE KERNELS NOT COMPARED: oracle/_ref/libref_kernels.so not built (needs /root/reference at build time)")]


def test_reference_kernels_are_present_when_they_were_built():
    """oracle/make_ref.sh leaves oracle/_ref/built.stamp next to the libraries: if the stamp travelled but a library did not load,
    the comparison below would silently not happen."""
    assert _HAVE_REF, "oracle/_ref/built.stamp (or GESPMM_REQUIRE_REF=1) says the reference kernels were built, but libref_kernels.so does not load"


WIDTHS = (1, 3, 16, 31, 32, 33, 41, 64, 100, 128, 200, 256, 512)


def _dev(G, val, B):
    rp = torch.from_numpy(np.ascontiguousarray(G["rowptr"])).cuda()
    ci = torch.from_numpy(np.ascontiguousarray(G["colind"])).cuda()
    v = torch.from_numpy(val).cuda() if val is not None else None
    return rp, ci, v, torch.from_numpy(B).cuda()


def _same(a, b, what):
    a, b = a.cpu().numpy(), b.cpu().numpy() if torch.is_tensor(b) else b
    if not np.array_equal(bits(a), bits(b)):
        bad = np.argwhere(bits(a) != bits(b))
        raise AssertionError("%s: %d/%d elements differ, first at %s: %r vs %r" %
                             (what, len(bad), a.size, tuple(bad[0]), a[tuple(bad[0])], b[tuple(bad[0])]))


@pytest.mark.parametrize("g", ("cora", "citeseer", "pubmed"))
def test_driver_kernels_spmmWrapper_all_methods(pkg, oracle, bundled, g):
    """spmmWrapper(method 0..4, tile_row 4 and 8) of the reference vs the product's variants and
    the oracle. tile_row 8 / method 2 is what the reference times (spmm_test.cu:756)."""
    from gespmm_amd import spmm

    G = bundled[g]
    val = oracle.hash_val(G["nnz"], seed=7)
    ones = np.ones(G["nnz"], dtype=np.float32)
    for N in WIDTHS:
        if g == "pubmed" and N in (200, 512):
            continue
        B = oracle.hash_B(G["K"], N, seed=1)
        for v, mode in ((ones, "golden"), (val, "fma")):
            rp, ci, vd, Bd = _dev(G, v, B)
            want = oracle.spmm(G["rowptr"], G["colind"], v, B, mode)
            mine = spmm.csr_spmm(rp, ci, vd, Bd)
            for method in range(5):
                for tile_row in (4, 8):
                    ref = ref_py.spmm_wrapper(method, tile_row, rp, ci, vd, Bd)
                    _same(ref, want, "%s N=%d reference method %d tile_row %d vs oracle(%s)" %
                          (g, N, method, tile_row, mode))
                    _same(mine, ref, "%s N=%d product vs reference method %d" % (g, N, method))
            for variant in (0, 1, 2, 3, 4):
                _same(spmm.csr_spmm(rp, ci, vd, Bd, variant=variant), want, "%s N=%d product variant %d" %
                      (g, N, variant))


@pytest.mark.parametrize("g", ("cora", "pubmed"))
def test_torch_op_kernels(pkg, oracle, bundled, g):
    """spmm_cuda / spmm_cuda_no_edge_value (pytorch-custom/spmm_kernel.cu) vs the product's
    csr_spmm / csr_spmm_no_edge_value — the three-way width dispatch included (N<32, <64, >=64)."""
    from gespmm_amd import spmm

    G = bundled[g]
    val = oracle.hash_val(G["nnz"], seed=7)
    for N in WIDTHS:
        if N < 2:
            continue  # 128/k rows per block with k = 1 exceeds nothing, but keep to the op's use
        B = oracle.hash_B(G["K"], N, seed=2)
        rp, ci, vd, Bd = _dev(G, val, B)
        _same(spmm.csr_spmm(rp, ci, vd, Bd), ref_py.spmm_cuda(rp, ci, vd, Bd), "%s N=%d valued op" % (g, N))
        _same(spmm.csr_spmm_no_edge_value(rp, ci, Bd), ref_py.spmm_cuda(rp, ci, None, Bd),
              "%s N=%d unweighted op" % (g, N))


def test_edge_shapes_against_reference_kernels(pkg, oracle):
    """Empty rows, rows around 32/64/128 entries, unsorted and repeated columns, K != M."""
    from gespmm_amd import spmm

    G = edge_case_csr(seed=2)
    val = oracle.hash_val(G["nnz"], seed=4)
    for N in (2, 5, 32, 41, 64, 96, 130):
        B = oracle.hash_B(G["K"], N, seed=N)
        rp, ci, vd, Bd = _dev(G, val, B)
        mine = spmm.csr_spmm(rp, ci, vd, Bd)
        for method in range(5):
            _same(mine, ref_py.spmm_wrapper(method, 4, rp, ci, vd, Bd), "edge N=%d method %d" % (N, method))
        _same(spmm.csr_spmm_no_edge_value(rp, ci, Bd), ref_py.spmm_cuda(rp, ci, None, Bd), "edge N=%d topo" % N)


def test_committed_valued_fma_vectors_are_the_reference_kernels_output(oracle, bundled):
    with open(os.path.join(GOLDEN, "spmm_checksums.json")) as f:
        sums = json.load(f)["graphs"]
    for g in ("cora", "citeseer", "pubmed"):
        G = bundled[g]
        val = oracle.hash_val(G["nnz"], seed=7)
        for N in (3, 16, 32, 41, 64, 128, 512):
            B = oracle.hash_B(G["K"], N, seed=1)
            rp, ci, vd, Bd = _dev(G, val, B)
            C = ref_py.spmm_wrapper(2, 8, rp, ci, vd, Bd).cpu().numpy()
            exp = sums[g][str(N)]["valued_fma"]
            assert int(np.bitwise_xor.reduce(bits(C).ravel())) == exp["xor"], (g, N)
            assert float(C.astype(np.float64).sum()) == exp["sum"], (g, N)
            for r, c, b in exp["samples"]:
                assert int(bits(C[r, c:c + 1])[0]) == b


def test_random_graph_full_width_sweep(pkg, oracle):
    """A seeded power-law graph (50 k rows, 600 k entries, hubs up to a few thousand entries)."""
    from gespmm_amd import graphs, spmm

    M = 50_000
    rp, ci = graphs.synthetic_csr(M, 600_000, symmetric=True, seed=11, device="cuda")
    nnz = int(ci.numel())
    val = torch.from_numpy(oracle.hash_val(nnz, seed=13)).cuda()
    for N in (32, 128, 256):
        B = torch.from_numpy(oracle.hash_B(M, N, seed=N)).cuda()
        ref = ref_py.spmm_wrapper(2, 8, rp, ci, val, B)
        _same(spmm.csr_spmm(rp, ci, val, B), ref, "power-law N=%d" % N)
        plan = spmm.SpmmPlan(rp, ci, M, N, values=val, reorder=True)
        _same(spmm.csr_spmm(rp, ci, val, B, plan=plan), ref, "power-law N=%d through a plan" % N)
