"""Soak of the staged-rows kernel: seeded random matrices (square / rectangular, local + far columns, empty rows, repeats, rows up to the 2048-entry
limit) through a forced staged plan against the plain call's strict-order bits, valued and unweighted, N = 128, 256 and (round 4: 256-column
tiles) 512 / 1024; every fifth seed also runs gespmm_plan_tune on an AUTO plan at N = 32 / 64 / 128 (any candidate may win: same bits).
    python scripts/staged_soak.py [first_seed] [count]"""
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
    """Seeds first .. first + count - 1; returns (#products compared, #plans that ran the staged-rows kernel). Raises AssertionError on the
    first product whose bits differ."""
    t0 = time.time()
    staged = checked = 0
    for seed in range(first, first + count):
        rng = np.random.RandomState(seed)
        M = int(rng.choice([1, 3, 63, 64, 65, 95, 96, 97, 127, 128, 129, 500, 2000, 9000, 40000]))
        K = M if rng.rand() < 0.6 else int(rng.randint(1, 20000))
        max_deg = int(rng.choice([1, 3, 17, 40, 120, 300]))
        rowptr, colind = _random_local_csr(rng, M, K, max_deg, local=int(rng.choice([1, 8, 60, 400])), p_empty=float(rng.choice([0.0, 0.1, 0.5])))
        if rng.rand() < 0.2 and M >= 64:
            r = int(rng.randint(0, M)); n_big = int(rng.choice([1000, 2047, 2048, 2049, 3000, 7000]))
            extra = rng.randint(0, K, size=n_big).astype(np.int32)
            d = n_big - (rowptr[r + 1] - rowptr[r])
            colind = np.concatenate([colind[:rowptr[r]], extra, colind[rowptr[r + 1]:]])
            rowptr = rowptr.copy(); rowptr[r + 1:] += d
        if colind.size == 0:
            continue
        rp, ci = torch.from_numpy(rowptr).cuda(), torch.from_numpy(colind).cuda()
        val = torch.from_numpy((rng.rand(colind.size).astype(np.float32) - 0.5)).cuda()
        for N in (128, 256, 512, 32) + ((1024, 16, 64) if seed % 7 == 0 else ()) + ((64,) if seed % 3 == 0 else ()):
            B = torch.from_numpy((rng.rand(K, N).astype(np.float32) - 0.5)).cuda()
            plan = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True, kernel="staged", flags=0x100)
            staged += "kernel=staged-rows" in plan.describe()
            want = spmm.csr_spmm(rp, ci, val, B, cfg={"flags": 0x100})
            got = spmm.csr_spmm(rp, ci, val, B, plan=plan)
            assert torch.equal(got.view(torch.int32), want.view(torch.int32)), ("valued", seed, M, K, N, max_deg, plan.describe())
            want = spmm.csr_spmm_no_edge_value(rp, ci, B, cfg={"flags": 0x100})
            got = spmm.csr_spmm_no_edge_value(rp, ci, B, plan=plan)
            assert torch.equal(got.view(torch.int32), want.view(torch.int32)), ("unweighted", seed, M, K, N, max_deg, plan.describe())
            checked += 2
            del plan
        if seed % 5 == 0:
            for N in (32, 64, 128):
                B = torch.from_numpy((rng.rand(K, N).astype(np.float32) - 0.5)).cuda()
                plan = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True, flags=0x100)
                want = spmm.csr_spmm(rp, ci, val, B, cfg={"flags": 0x100})
                got = plan.tune(B, reps=1)
                assert torch.equal(got.view(torch.int32), want.view(torch.int32)), ("tune", seed, M, K, N, plan.describe())
                got = spmm.csr_spmm(rp, ci, val, B, plan=plan)
                assert torch.equal(got.view(torch.int32), want.view(torch.int32)), ("after tune", seed, M, K, N, plan.describe())
                checked += 2
                del plan
    if verbose:
        print("staged soak: seeds %d..%d, %d products compared bit for bit (%d plans on the staged-rows kernel), %.0f s: all equal"
              % (first, first + count - 1, checked, staged, time.time() - t0))
    return checked, staged


if __name__ == "__main__":
    soak(int(sys.argv[1]) if len(sys.argv) > 1 else 5000, int(sys.argv[2]) if len(sys.argv) > 2 else 300)
