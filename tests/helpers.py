"""Shared helpers for the test-suite (plain module: tests/ is on sys.path)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def edge_case_csr(seed=0):
    """Hand-shaped CSR the reference never tests (SURVEY.md §8 c3 iii): empty rows, rows
    of exactly 1/63/64/65/127/128/129/200 entries, an empty tail, M not a multiple of
    any block size, rectangular (K != M), unsorted columns and repeated columns."""
    rng = np.random.RandomState(seed)
    K = 301
    degs = [0, 1, 63, 64, 65, 0, 0, 127, 128, 129, 200, 2, 3, 0, 5, 31, 32, 33, 7, 0, 0]
    rowptr = np.zeros(len(degs) + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(degs)
    colind = rng.randint(0, K, size=int(rowptr[-1])).astype(np.int32)  # unsorted, with repeats
    return {"M": len(degs), "K": K, "nnz": int(rowptr[-1]), "rowptr": rowptr, "colind": colind}


def sampled_rows_equal_oracle(oracle, rowptr, colind, val, B, C, nrows=512, seed=0):
    """`nrows` sampled rows of C = A @ B (device tensors) against the oracle's device-arithmetic chain on the extracted
    sub-matrix, bit for bit. Only the B rows those CSR rows touch travel to the host."""
    import torch

    M = rowptr.numel() - 1
    rng = np.random.RandomState(seed)
    rows = np.sort(rng.choice(M, min(nrows, M), replace=False))
    rph = rowptr.cpu().numpy()
    sub_ptr = np.zeros(len(rows) + 1, dtype=np.int32)
    sub_ptr[1:] = np.cumsum(rph[rows + 1] - rph[rows])
    sel = torch.from_numpy(np.concatenate([np.arange(rph[r], rph[r + 1]) for r in rows]).astype(np.int64)).to(colind.device)
    cih = colind[sel].cpu().numpy()
    vh = val[sel].cpu().numpy() if val is not None else None
    cols_u, inv = np.unique(cih, return_inverse=True)
    Bsub = B[torch.from_numpy(cols_u.astype(np.int64)).to(B.device)].cpu().numpy()
    ref = oracle.spmm(sub_ptr, inv.astype(np.int32), vh, Bsub, "fma" if vh is not None else "golden")
    got = C[torch.from_numpy(rows).to(C.device)].cpu().numpy()
    return bool(np.array_equal(bits(got), bits(ref)))
