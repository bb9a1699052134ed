"""Soak of the general staged-rows kernel (csrc/spmm_staged_gen.hip): seeded random matrices (the generator of scripts/staged_soak.py) through a
forced staged plan at random widths 1 .. 700 (every lane vector: odd, 2 mod 4, 0 mod 4; one to eleven column tiles) against the plain call's
strict-order bits — valued, unweighted and the max reducer.
    python scripts/staged_gen_soak.py [first_seed] [count]"""
import os, sys, time
import numpy as np
import torch
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import gespmm_amd
from gespmm_amd import spmm
from test_gpu_plan_staged import _random_local_csr

TUNED = (16, 32, 64, 128, 256, 512, 1024)


def soak(first, count, verbose=True):
    t0 = time.time()
    staged = checked = 0
    for seed in range(first, first + count):
        rng = np.random.RandomState(seed)
        M = int(rng.choice([1, 3, 63, 64, 65, 95, 96, 97, 127, 128, 129, 191, 192, 193, 500, 2000, 9000, 40000]))
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
        widths = [int(rng.randint(1, 65)), int(rng.randint(65, 260)), int(rng.choice([2, 4])) * int(rng.randint(17, 176))]
        if seed % 4 == 0:
            widths.append(int(rng.choice(TUNED[3:])))  # (the tuned widths: the max reducer walks their tables through the general kernel)
        for N in widths:
            if K * N * 4 >= (1 << 31) or M * N * 4 >= (1 << 31):
                continue
            B = torch.from_numpy((rng.rand(K, N).astype(np.float32) - 0.5)).cuda()
            if N not in TUNED:
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
            if N not in TUNED[:3]:
                plan = spmm.SpmmPlan(rp, ci, K, N, reorder=True, kernel="staged", flags=0x100)
                staged += "kernel=staged-rows" in plan.describe()
                empty = float(rng.choice([-10000.0, -0.25]))
                want = spmm.csr_spmm_max(rp, ci, B, empty_value=empty)
                got = plan.run(None, B, reduce_max=empty)
                assert torch.equal(got.view(torch.int32), want.view(torch.int32)), ("max", seed, M, K, N, max_deg, plan.describe())
                checked += 1
                del plan
    if verbose:
        print("general staged soak: seeds %d..%d, %d products compared bit for bit (%d plans on the staged-rows kernel), %.0f s: all equal"
              % (first, first + count - 1, checked, staged, time.time() - t0))
    return checked, staged


if __name__ == "__main__":
    soak(int(sys.argv[1]) if len(sys.argv) > 1 else 7000, int(sys.argv[2]) if len(sys.argv) > 2 else 300)
