"""The oracle's SpMM / SDDMM / csr2csc restatements: committed regression vectors,
self-consistency between the reference's two CPU statements (gather form of
spmm_test.cu:595-605, scatter form of gunrock's CPU_Reference), and an independent
float64 check with scipy."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

from helpers import GOLDEN, bits, edge_case_csr
from golden.make_golden import N_LIST, sample_positions


@pytest.fixture(scope="module")
def checksums():
    with open(os.path.join(GOLDEN, "spmm_checksums.json")) as f:
        return json.load(f)["graphs"]


@pytest.mark.parametrize("g", ("cora", "citeseer", "pubmed"))
def test_committed_vectors(oracle, bundled, checksums, g):
    G = bundled[g]
    val = oracle.hash_val(G["nnz"], seed=7)
    for N in N_LIST:
        if g == "pubmed" and N == 512:
            continue  # 90 MFLOP through the literal loop: covered by cora/citeseer
        B = oracle.hash_B(G["K"], N, seed=1)
        for mode, v in (("unweighted_golden", None), ("valued_golden", val), ("valued_fma", val)):
            C = oracle.spmm(G["rowptr"], G["colind"], v, B, mode="fma" if mode == "valued_fma" else "golden")
            exp = checksums[g][str(N)][mode]
            assert int(np.bitwise_xor.reduce(bits(C).ravel())) == exp["xor"], (g, N, mode)
            assert float(C.astype(np.float64).sum()) == exp["sum"]
            for r, c, b in exp["samples"]:
                assert int(bits(C[r, c:c + 1])[0]) == b


@pytest.mark.parametrize("g", ("cora", "citeseer"))
def test_golden_equals_fma_and_scatter_when_unweighted(oracle, bundled, g):
    """A == 1 makes every product exact, so the CPU golden (mul+add), the device
    arithmetic (fma) and gunrock's scatter form must agree bit for bit."""
    G = bundled[g]
    B = oracle.hash_B(G["K"], 41, seed=5)
    a = oracle.spmm(G["rowptr"], G["colind"], None, B, "golden")
    assert np.array_equal(bits(a), bits(oracle.spmm(G["rowptr"], G["colind"], None, B, "fma")))
    assert np.array_equal(bits(a), bits(oracle.spmm_scatter(G["rowptr"], G["colind"], B)))
    assert np.array_equal(bits(a), bits(oracle.spmm(G["rowptr"], G["colind"], None, B, "omp")))
    ones = np.ones(G["nnz"], np.float32)
    assert np.array_equal(bits(a), bits(oracle.spmm(G["rowptr"], G["colind"], ones, B, "fma")))


@pytest.mark.parametrize("valued", (False, True))
def test_against_scipy_float64(oracle, bundled, valued):
    G = bundled["pubmed"]
    N = 32
    B = oracle.hash_B(G["K"], N, seed=2)
    val = oracle.hash_val(G["nnz"], 9) if valued else None
    A = sp.csr_matrix((val.astype(np.float64) if valued else np.ones(G["nnz"]), G["colind"], G["rowptr"]),
                      shape=(G["M"], G["K"]))
    ref = A @ B.astype(np.float64)
    scale = oracle.spmm_abs(G["rowptr"], G["colind"], val, B)
    for mode in ("golden", "fma"):
        C = oracle.spmm(G["rowptr"], G["colind"], val, B, mode)
        # tolerance of north_star (1e-4 relative), scaled as SURVEY.md §8 c4 prescribes
        assert np.all(np.abs(C - ref) <= 1e-4 * np.maximum(np.abs(ref), scale) + 1e-30)
        assert np.abs(C - ref).max() < 2e-5


def test_edge_shapes(oracle):
    G = edge_case_csr()
    B = oracle.hash_B(G["K"], 7, seed=3)
    C = oracle.spmm(G["rowptr"], G["colind"], None, B, "golden")
    for r in range(G["M"]):
        cols = G["colind"][G["rowptr"][r]:G["rowptr"][r + 1]]
        if len(cols) == 0:
            assert np.all(C[r] == 0)
    A = sp.csr_matrix((np.ones(G["nnz"]), G["colind"], G["rowptr"]), shape=(G["M"], G["K"]))
    assert np.abs(A @ B.astype(np.float64) - C).max() < 1e-4


def test_max_reducer(oracle):
    G = edge_case_csr(1)
    B = oracle.hash_B(G["K"], 5, seed=4)
    C = oracle.spmm_max(G["rowptr"], G["colind"], B, init=-10000.0)
    for r in range(G["M"]):
        cols = G["colind"][G["rowptr"][r]:G["rowptr"][r + 1]]
        exp = np.maximum(B[cols].max(axis=0), -10000.0) if len(cols) else np.full(5, -10000.0, np.float32)
        assert np.array_equal(C[r], exp.astype(np.float32))


def test_sddmm(oracle):
    G = edge_case_csr(2)
    N = 19
    D1 = oracle.hash_B(G["M"], N, seed=6)
    D2 = oracle.hash_B(G["K"], N, seed=7)
    rows = np.repeat(np.arange(G["M"], dtype=np.int32), np.diff(G["rowptr"]))
    out_coo, scale = oracle.sddmm(rows, G["colind"], D1, D2, csr=False)
    out_csr, _ = oracle.sddmm(G["rowptr"], G["colind"], D1, D2, csr=True)
    ref = np.einsum("ej,ej->e", D1[rows].astype(np.float64), D2[G["colind"]].astype(np.float64))
    assert np.array_equal(out_coo, ref.astype(np.float32))
    assert np.array_equal(out_coo, out_csr), "CSR row search (findRow) must agree with explicit rows"
    assert np.all(scale >= np.abs(ref) - 1e-12)


def test_csr2csc(oracle):
    G = edge_case_csr(3)
    val = oracle.hash_val(G["nnz"], 1)
    colptr, rowind, cv = oracle.csr2csc(G["M"], G["K"], G["rowptr"], G["colind"], val)
    A = sp.csr_matrix((val, G["colind"], G["rowptr"]), shape=(G["M"], G["K"]))
    assert np.array_equal(colptr, np.concatenate([[0], np.cumsum(np.bincount(G["colind"], minlength=G["K"]))]))
    for c in range(G["K"]):
        seg = rowind[colptr[c]:colptr[c + 1]]
        assert np.all(np.diff(seg) >= 0), "rows ascending inside a column"
    T = sp.csr_matrix((cv, rowind, colptr), shape=(G["K"], G["M"]))
    assert np.abs((A.T - T)).max() < 1e-6


def test_fill_B_value_set(oracle):
    B = oracle.fill_B_rand(1, 50, 40)
    H = oracle.hash_B(50, 40, 1)
    for X in (B, H):
        q = np.round(X * 100).astype(int)
        assert q.min() >= -50 and q.max() <= 49
        assert np.array_equal((q.astype(np.float32) / np.float32(100)), X)
    assert len(sample_positions(10, 10)) == 64
