"""Seeded random sweep over shapes, degree laws, widths, variants and launch knobs: whatever
the selection logic picks (kernel generation, task size, slab blocking, long-row pass, 32/64-bit
offsets), variants 0-4 must reproduce the oracle's fp32 chain bit for bit; only rows the long-row
pass re-associates are held to the 1e-4 tolerance instead."""
import numpy as np
import pytest
import torch

from helpers import bits

pytestmark = pytest.mark.gpu


def random_csr(rng):
    M = int(rng.choice([1, 2, 7, 63, 64, 65, 300, 1000, 2500]))
    K = int(rng.choice([1, 5, 64, 333, 1000, 4000]))
    law = rng.choice(["uniform", "powerlaw", "sparse", "hub", "empty"])
    if law == "uniform":
        degs = rng.randint(0, 40, size=M)
    elif law == "powerlaw":
        degs = np.minimum((rng.pareto(1.2, size=M) * 3).astype(np.int64), 3000)
    elif law == "sparse":
        degs = (rng.rand(M) < 0.2).astype(np.int64) * rng.randint(1, 4, size=M)
    elif law == "hub":
        degs = rng.randint(0, 6, size=M)
        degs[rng.randint(0, M)] = int(rng.choice([2049, 4097, 9000]))
    else:
        degs = np.zeros(M, dtype=np.int64)
    rowptr = np.zeros(M + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(degs)
    colind = rng.randint(0, K, size=int(rowptr[-1])).astype(np.int32)
    if rng.rand() < 0.5:  # ascending columns inside every row, as a loader would produce
        for r in range(M):
            colind[rowptr[r]:rowptr[r + 1]].sort()
    return {"M": M, "K": K, "nnz": int(rowptr[-1]), "rowptr": rowptr, "colind": colind}, law


def test_random_shapes_and_knobs(pkg, oracle):
    from gespmm_amd import _lib, spmm

    knobs = [0, _lib.FLAG_SEG_STREAM, _lib.FLAG_BATCH_STREAM, _lib.FLAG_NT_STORE, _lib.FLAG_FORCE_IDX64,
             _lib.FLAG_NO_XCD_REMAP, _lib.FLAG_SHALLOW_UNROLL, _lib.FLAG_SLAB_BLOCKED, _lib.FLAG_SPLIT_LONG_ROWS,
             _lib.FLAG_SPLIT_LONG_ROWS | _lib.FLAG_STRICT_ORDER, _lib.FLAG_SLAB_BLOCKED | _lib.FLAG_FORCE_IDX64]
    rng = np.random.RandomState(20260928)
    for case in range(600):
        G, law = random_csr(rng)
        N = int(rng.choice([1, 2, 3, 4, 7, 16, 31, 32, 33, 64, 65, 100, 127, 128, 129, 130, 200, 256, 258, 260, 384, 512, 513]))
        variant = int(rng.choice([-1, 0, 1, 2, 3, 4]))
        flags = int(knobs[rng.randint(len(knobs))])
        cfg = {"flags": flags, "rows_per_wave": int(rng.choice([0, 0, 1, 2, 8, 32])),
               "slab_rows": int(rng.choice([0, 1, 17, 500]))}
        valued = bool(rng.rand() < 0.6)
        val = oracle.hash_val(G["nnz"], seed=case) if valued else None
        B = oracle.hash_B(G["K"], N, seed=case + 1)
        ref = oracle.spmm(G["rowptr"], G["colind"], val, B, "fma")
        rp = torch.from_numpy(G["rowptr"]).cuda()
        ci = torch.from_numpy(G["colind"]).cuda()
        Bd = torch.from_numpy(B).cuda()
        if valued:
            C = spmm.csr_spmm(rp, ci, torch.from_numpy(val).cuda(), Bd, variant=variant, cfg=cfg)
        else:
            C = spmm.csr_spmm_no_edge_value(rp, ci, Bd, variant=variant, cfg=cfg)
        C = C.cpu().numpy()
        what = "case %d: %s M=%d K=%d nnz=%d N=%d variant=%d cfg=%r valued=%r" % (
            case, law, G["M"], G["K"], G["nnz"], N, variant, cfg, valued)
        split = (flags & _lib.FLAG_SPLIT_LONG_ROWS) and not (flags & _lib.FLAG_STRICT_ORDER) and variant != 0
        degs = np.diff(G["rowptr"])
        exact_rows = np.nonzero(degs <= 2048)[0] if split else np.arange(G["M"])
        assert np.array_equal(bits(C[exact_rows]), bits(ref[exact_rows])), what
        if split and len(exact_rows) < G["M"]:
            long_rows = np.nonzero(degs > 2048)[0]
            scale = oracle.spmm_abs(G["rowptr"], G["colind"], val, B)
            tol = 1e-4 * np.maximum(np.abs(ref[long_rows]), scale[long_rows])
            assert np.all(np.abs(C[long_rows].astype(np.float64) - ref[long_rows]) <= tol + 1e-30), what


def test_wide_outputs_use_several_column_tiles(pkg, oracle):
    """N beyond one column tile (512 columns at V=4, S=2): the tile index joins the work-item id."""
    from gespmm_amd import _lib, spmm

    rng = np.random.RandomState(7)
    for N in (513, 777, 1024, 1500, 2048):
        G, _ = random_csr(rng)
        val = oracle.hash_val(G["nnz"], seed=N)
        B = oracle.hash_B(G["K"], N, seed=N)
        ref = oracle.spmm(G["rowptr"], G["colind"], val, B, "fma")
        rp, ci = torch.from_numpy(G["rowptr"]).cuda(), torch.from_numpy(G["colind"]).cuda()
        vd, Bd = torch.from_numpy(val).cuda(), torch.from_numpy(B).cuda()
        for variant in (-1, 0, 1, 3, 4, 5):
            for flags in (0, _lib.FLAG_SEG_STREAM, _lib.FLAG_SLAB_BLOCKED, _lib.FLAG_NO_XCD_REMAP):
                if variant in (0, 5) and flags:
                    continue
                C = spmm.csr_spmm(rp, ci, vd, Bd, variant=variant, cfg={"flags": flags, "slab_rows": 300}).cpu().numpy()
                what = "N=%d variant=%d flags=%#x M=%d nnz=%d" % (N, variant, flags, G["M"], G["nnz"])
                if variant == 5:
                    scale = oracle.spmm_abs(G["rowptr"], G["colind"], val, B)
                    assert np.all(np.abs(C.astype(np.float64) - ref) <= 1e-4 * np.maximum(np.abs(ref), scale) + 1e-30), what
                else:
                    assert np.array_equal(bits(C), bits(ref)), what
