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
