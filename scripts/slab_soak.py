#!/usr/bin/env python3
"""Soak of the column-slab tables (plan.cpp build_slab_tables + the continuing launches of spmm_staged.hip): seeded random community matrices —
sizes around the block height, K != M, empty rows, repeated columns, mean degree 20 ... 900 (2 ... 14 column ranges by the rule), hub rows up
to and beyond the per-range row limit — through kernel="staged-slabs" against the plain strict-order call, bit for bit: valued, unweighted,
new values, C filled with NaN before the call.     python scripts/slab_soak.py [first_seed] [count]"""
import os, sys, time
import numpy as np
import torch
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
import gespmm_amd
from gespmm_amd import _lib, spmm


def community_csr(rng, M, K, comm, deg_in, deg_out, p_empty):
    ncomm = (M + comm - 1) // comm
    cols_of = [rng.choice(K, size=min(K, max(8, int(rng.choice([2, 6, 12])) * comm)), replace=True) for _ in range(ncomm)]
    shuffle = rng.permutation(M)
    rows = []
    for i in range(M):
        if rng.rand() < p_empty:
            rows.append(np.zeros(0, dtype=np.int32)); continue
        a = rng.choice(cols_of[shuffle[i] // comm], size=rng.randint(1, 2 * deg_in + 1))
        b = rng.randint(0, K, size=rng.randint(0, 2 * deg_out + 1))
        rows.append(np.sort(np.concatenate([a, b]).astype(np.int32), kind="stable"))
    return rows


def soak(first, count):
    t0 = time.time()
    slabbed = checked = 0
    for seed in range(first, first + count):
        rng = np.random.RandomState(seed)
        M = int(rng.choice([1, 5, 95, 96, 97, 191, 193, 700, 2500, 6000]))
        K = M if rng.rand() < 0.5 else int(rng.randint(1, 9000))
        deg = int(rng.choice([10, 40, 70, 150, 300, 450]))
        rows = community_csr(rng, M, K, comm=int(rng.choice([40, 150, 400])), deg_in=deg, deg_out=max(1, deg // 3),
                             p_empty=float(rng.choice([0.0, 0.05, 0.4])))
        if rng.rand() < 0.25 and M >= 96:  # a hub row: around the per-range limit of 2048 entries times a few ranges
            r = int(rng.randint(0, M))
            rows[r] = np.sort(rng.randint(0, K, size=int(rng.choice([2047, 2049, 5000, 12000, 30000]))).astype(np.int32))
        rowptr = np.zeros(M + 1, dtype=np.int32)
        rowptr[1:] = np.cumsum([len(r) for r in rows])
        colind = np.concatenate(rows).astype(np.int32) if rowptr[-1] else np.zeros(0, dtype=np.int32)
        nnz = int(colind.size)
        if nnz == 0:
            continue
        rp, ci = torch.from_numpy(rowptr).cuda(), torch.from_numpy(colind).cuda()
        val = torch.rand(nnz, device="cuda") - 0.5
        B = torch.rand(K, 128, device="cuda") - 0.5
        plan = spmm.SpmmPlan(rp, ci, K, 128, values=val, reorder=True, kernel="staged-slabs")
        slabbed += "kernel=staged-slabs" in plan.describe()
        strict = {"flags": _lib.FLAG_STRICT_ORDER}
        for v in (val, None, torch.rand(nnz, device="cuda") - 0.5):
            C = torch.full((M, 128), float("nan"), device="cuda")
            if v is None:
                spmm.csr_spmm_no_edge_value(rp, ci, B, out=C, plan=plan)
                ref = spmm.csr_spmm_no_edge_value(rp, ci, B, cfg=strict)
            else:
                spmm.csr_spmm(rp, ci, v, B, out=C, plan=plan)
                ref = spmm.csr_spmm(rp, ci, v, B, cfg=strict)
            if "kernel=staged-slabs" in plan.describe():
                assert torch.equal(C.view(torch.int32), ref.view(torch.int32)), (seed, M, K, deg, plan.describe()[-200:])
                checked += 1
            else:  # (a hub row beyond the per-range limit: the plan's streaming kernels, whose long-row pass re-associates — tolerance class)
                va = torch.ones(nnz, device="cuda") if v is None else v.abs()
                scale = torch.maximum(ref.abs(), spmm.csr_spmm(rp, ci, va, B.abs(), cfg=strict))  # sum |a b| per element (the tolerance class)
                assert bool(((C - ref).abs() <= 1e-4 * scale + 1e-30).all()), (seed, M, K, deg, plan.describe()[-200:])
        del plan
    print("column-slab soak: seeds %d..%d, %d products compared bit for bit (%d plans on the slab tables), %.0f s: all equal" % (
        first, first + count - 1, checked, slabbed, time.time() - t0))


if __name__ == "__main__":
    soak(int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 100)
