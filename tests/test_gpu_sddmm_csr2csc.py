"""SDDMM and CSR->CSC through the C ABI against the oracle."""
import numpy as np
import pytest
import torch

from helpers import edge_case_csr

pytestmark = pytest.mark.gpu


def _rows_of(G):
    return np.repeat(np.arange(G["M"], dtype=np.int32), np.diff(G["rowptr"]))


@pytest.mark.parametrize("which", ("edge", "cora", "pubmed"))
def test_sddmm_coo_and_csr(pkg, oracle, bundled, which):
    from gespmm_amd import sddmm

    G = edge_case_csr(2) if which == "edge" else bundled[which]
    rows = _rows_of(G)
    for N in (1, 2, 3, 4, 6, 32, 41, 100, 128, 256):
        if which == "pubmed" and N not in (3, 32, 100, 128):
            continue  # SURVEY.md §8 c3(iv) widths
        D1 = oracle.hash_B(G["M"], N, seed=N)
        D2 = oracle.hash_B(G["K"], N, seed=N + 1)
        ref, scale = oracle.sddmm(rows, G["colind"], D1, D2, csr=False)
        d1, d2 = torch.from_numpy(D1).cuda(), torch.from_numpy(D2).cuda()
        ci = torch.from_numpy(G["colind"]).cuda()
        out_coo = sddmm.coo_sddmm(torch.from_numpy(rows).cuda(), ci, d1, d2).cpu().numpy()
        out_csr = sddmm.csr_sddmm(torch.from_numpy(G["rowptr"]).cuda(), ci, d1, d2).cpu().numpy()
        # tolerance of north_star (1e-4 relative), scaled by sum|d1*d2|
        tol = 1e-4 * np.maximum(np.abs(ref), scale) + 1e-30
        assert np.all(np.abs(out_coo.astype(np.float64) - ref) <= tol), (which, N)
        assert np.array_equal(out_coo, out_csr), "COO and CSR forms must agree exactly"


def test_sddmm_empty_and_errors(pkg):
    from gespmm_amd import sddmm

    e = torch.zeros(0, dtype=torch.int32, device="cuda")
    D = torch.ones(4, 8, device="cuda")
    assert sddmm.coo_sddmm(e, e, D, D).numel() == 0
    with pytest.raises(ValueError):
        sddmm.coo_sddmm(e, e, D, torch.ones(4, 7, device="cuda"))
    with pytest.raises(ValueError):
        sddmm.csr_sddmm(torch.zeros(3, dtype=torch.int32, device="cuda"), e, D, D)


@pytest.mark.parametrize("which", ("edge", "citeseer", "pubmed"))
def test_csr2csc(pkg, oracle, bundled, which):
    from gespmm_amd import spmm

    G = edge_case_csr(3) if which == "edge" else bundled[which]
    val = oracle.hash_val(G["nnz"], seed=1)
    colptr_ref, rowind_ref, cv_ref = oracle.csr2csc(G["M"], G["K"], G["rowptr"], G["colind"], val)
    colptr = torch.full((G["K"] + 1,), -7, dtype=torch.int32, device="cuda")
    rowind = torch.full((G["nnz"],), -7, dtype=torch.int32, device="cuda")
    cv = spmm.csr2csc(torch.from_numpy(G["rowptr"]).cuda(), torch.from_numpy(G["colind"]).cuda(), colptr, rowind,
                      torch.from_numpy(val).cuda())
    assert np.array_equal(colptr.cpu().numpy(), colptr_ref)
    assert np.array_equal(rowind.cpu().numpy(), rowind_ref), "stable: rows ascending inside each column"
    assert np.array_equal(cv.cpu().numpy(), cv_ref)


def test_csr2csc_round_trip_and_backward_operand(pkg, oracle, bundled):
    """CSC of the CSC is the CSR again (columns sorted inside rows), and SpMM on the CSC
    arrays is A^T @ X — the operand op.py's backward needs."""
    from gespmm_amd import spmm

    G = bundled["cora"]
    rp = torch.from_numpy(G["rowptr"]).cuda()
    ci = torch.from_numpy(G["colind"]).cuda()
    val = torch.from_numpy(oracle.hash_val(G["nnz"], seed=3)).cuda()
    colptr = torch.empty(G["K"] + 1, dtype=torch.int32, device="cuda")
    rowind = torch.empty(G["nnz"], dtype=torch.int32, device="cuda")
    cv = spmm.csr2csc(rp, ci, colptr, rowind, val)
    rp2 = torch.empty(G["M"] + 1, dtype=torch.int32, device="cuda")
    ci2 = torch.empty(G["nnz"], dtype=torch.int32, device="cuda")
    v2 = spmm.csr2csc(colptr, rowind, rp2, ci2, cv)
    assert torch.equal(rp2, rp) and torch.equal(ci2, ci) and torch.equal(v2, val)
    X = torch.from_numpy(oracle.hash_B(G["M"], 16, seed=9)).cuda()
    At_X = spmm.csr_spmm(colptr, rowind, cv, X).cpu().numpy()
    import scipy.sparse as sp

    A = sp.csr_matrix((val.cpu().numpy().astype(np.float64), G["colind"], G["rowptr"]), shape=(G["M"], G["K"]))
    assert np.abs(A.T @ X.cpu().numpy().astype(np.float64) - At_X).max() < 1e-5


def test_sddmm_cache_blocked_form_matches_streaming_forms(pkg, oracle):
    """Dense patterns (mean degree >= 64, D2 several slabs large) take the cache-blocked CSR kernel:
    one launch per column slab. Same per-edge arithmetic, so CSR == COO bit for bit, sorted or not."""
    from gespmm_amd import sddmm

    rng = np.random.RandomState(3)
    M = 30000
    deg = rng.randint(60, 120, size=M)
    deg[::1000] = 0
    deg[7] = 5000
    rowptr = np.zeros(M + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(deg)
    nnz = int(rowptr[-1])
    colind = rng.randint(0, M, size=nnz).astype(np.int32)
    for r in range(0, M, 2):  # every other row ascending, the rest in file order
        colind[rowptr[r]:rowptr[r + 1]].sort()
    rows = np.repeat(np.arange(M, dtype=np.int32), deg)
    rp, ci, ri = (torch.from_numpy(x).cuda() for x in (rowptr, colind, rows))
    for N in (128, 100, 65, 256):
        D1 = torch.rand(M, N, device="cuda") - 0.5
        D2 = torch.rand(M, N, device="cuda") - 0.5
        o_csr = sddmm.csr_sddmm(rp, ci, D1, D2)
        o_coo = sddmm.coo_sddmm(ri, ci, D1, D2)
        assert torch.equal(o_csr, o_coo), N
        idx = torch.from_numpy(rng.randint(0, nnz, size=20000)).cuda()
        ref = (D1[ri[idx].long()].double() * D2[ci[idx].long()].double()).sum(1)
        scale = (D1[ri[idx].long()].double() * D2[ci[idx].long()].double()).abs().sum(1)
        assert torch.all((o_csr[idx].double() - ref).abs() <= 1e-4 * torch.maximum(ref.abs(), scale) + 1e-30), N
