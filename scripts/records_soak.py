"""Soak of the padded-record kernel (csrc/spmm_records.hip): seeded random matrices (the generator of scripts/staged_soak.py: empty rows,
rows of 1 .. 300 entries, local and scattered columns, K != M, now and then one row of 700 .. 1024 entries — and one beyond the limit, which
must fall back) through a forced records plan at random widths 4 .. 64 (lane groups of 4 / 8 / 16; multiples of 4 and not), clustered and storage
order, random task lengths, against the plain call's strict-order bits — valued, unweighted, and after new values.
    python scripts/records_soak.py [first_seed] [count]"""
import os, sys, time
import numpy as np
import torch
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import gespmm_amd
from gespmm_amd import spmm
from test_gpu_plan_staged import _random_local_csr


def soak(first, count, verbose=True):
    t0 = time.time()
    served = checked = 0
    for seed in range(first, first + count):
        rng = np.random.RandomState(seed)
        M = int(rng.choice([1, 3, 15, 16, 17, 255, 256, 257, 511, 513, 2000, 9000, 40000]))
        K = M if rng.rand() < 0.6 else int(rng.randint(1, 20000))
        max_deg = int(rng.choice([1, 3, 8, 9, 17, 40, 120, 300]))
        rowptr, colind = _random_local_csr(rng, M, K, max_deg, local=int(rng.choice([1, 8, 60, 400])), p_empty=float(rng.choice([0.0, 0.1, 0.5])))
        big = 0
        if rng.rand() < 0.25 and M >= 16:
            r = int(rng.randint(0, M)); big = int(rng.choice([700, 1023, 1024, 1025, 3000]))
            extra = rng.randint(0, K, size=big).astype(np.int32)
            d = big - (rowptr[r + 1] - rowptr[r])
            colind = np.concatenate([colind[:rowptr[r]], extra, colind[rowptr[r + 1]:]])
            rowptr = rowptr.copy(); rowptr[r + 1:] += d
        if colind.size == 0:
            continue
        rp, ci = torch.from_numpy(rowptr).cuda(), torch.from_numpy(colind).cuda()
        val = torch.from_numpy((rng.rand(colind.size).astype(np.float32) - 0.5)).cuda()
        val2 = torch.from_numpy((rng.rand(colind.size).astype(np.float32) - 0.5)).cuda()
        for N in (4 * int(rng.randint(1, 5)), 4 * int(rng.randint(5, 9)), 4 * int(rng.randint(9, 17)), int(rng.randint(4, 65))):
            B = torch.from_numpy((rng.rand(K, N).astype(np.float32) - 0.5)).cuda()
            reorder = bool(rng.rand() < 0.7)
            plan = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=reorder, kernel="records", flags=0x100)
            on = "kernel=padded-records" in plan.describe()
            assert on == (big <= 1024 and int(np.diff(rowptr).max()) <= 1024), (seed, big, plan.describe())
            served += on
            for v in (val, None, val2):
                want = spmm.csr_spmm(rp, ci, v, B, cfg={"flags": 0x100}) if v is not None else spmm.csr_spmm_no_edge_value(rp, ci, B, cfg={"flags": 0x100})
                got = spmm.csr_spmm(rp, ci, v, B, plan=plan) if v is not None else spmm.csr_spmm_no_edge_value(rp, ci, B, plan=plan)
                assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (seed, M, K, N, max_deg, reorder, v is None, plan.describe())
                checked += 1
            del plan
    if verbose:
        print("padded-record soak: seeds %d..%d, %d products compared bit for bit (%d plans on the record kernel), %.0f s: all equal"
              % (first, first + count - 1, checked, served, time.time() - t0))
    return checked, served


if __name__ == "__main__":
    soak(int(sys.argv[1]) if len(sys.argv) > 1 else 9000, int(sys.argv[2]) if len(sys.argv) > 2 else 300)
