#!/usr/bin/env python3
"""One-off soak: many random matrices through every launch form (plain AUTO, explicit kernels / flags, plans with every kernel,
max reducer, unweighted), all compared bit for bit with the plain strict-order call; a sample also against the CPU oracle.

    python scripts/soak_fuzz.py [--cases 1500] [--seed 1]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=1500)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import numpy as np
    import torch

    import gespmm_amd  # noqa: F401
    from gespmm_amd import _lib, spmm

    rng = np.random.RandomState(args.seed)
    t0 = time.time()
    nplans = nflags = 0
    for case in range(args.cases):
        shape = rng.randint(0, 6)
        if shape == 0:    # small
            M, K = int(rng.randint(1, 400)), int(rng.randint(1, 400))
            deg = rng.geometric(0.3, size=M) - 1
        elif shape == 1:  # medium, short rows, many empty
            M, K = int(rng.randint(1000, 60000)), int(rng.randint(50, 60000))
            deg = rng.binomial(3, 0.4, size=M)
        elif shape == 2:  # medium, power-law-ish with hubs
            M, K = int(rng.randint(500, 20000)), int(rng.randint(500, 20000))
            deg = np.minimum((rng.pareto(1.2, size=M) * 3).astype(np.int64), 6000)
        elif shape == 3:  # long rows (slab / long-row territory when forced)
            M, K = int(rng.randint(50, 2000)), int(rng.randint(2000, 30000))
            deg = rng.randint(0, 700, size=M)
        elif shape == 4:  # rectangular, tiny K (every gather hits)
            M, K = int(rng.randint(1000, 40000)), int(rng.randint(1, 64))
            deg = rng.randint(0, 20, size=M)
        else:             # exactly tile-sized rows
            M, K = int(rng.randint(10, 3000)), int(rng.randint(100, 5000))
            deg = rng.choice([0, 1, 7, 8, 9, 31, 32, 33, 63, 64, 65, 127, 128, 129], size=M)
        rowptr = np.zeros(M + 1, dtype=np.int64)
        rowptr[1:] = np.cumsum(deg)
        nnz = int(rowptr[-1])
        if nnz >= 2**24:
            continue
        colind = rng.randint(0, K, size=nnz).astype(np.int32)
        if rng.rand() < 0.5 and nnz:  # sorted rows half of the time
            order = np.lexsort((colind, np.repeat(np.arange(M), deg)))
            colind = colind[order]
        N = int(rng.choice([1, 2, 3, 4, 8, 16, 31, 32, 33, 48, 64, 65, 100, 128, 130, 192, 256, 260, 512]))
        rp = torch.from_numpy(rowptr.astype(np.int32)).cuda()
        ci = torch.from_numpy(colind).cuda()
        val = torch.rand(nnz, device="cuda") - 0.5
        B = torch.rand(K, N, device="cuda") - 0.5
        strict = _lib.FLAG_STRICT_ORDER
        ref = spmm.csr_spmm(rp, ci, val, B, cfg={"flags": strict})
        refu = spmm.csr_spmm_no_edge_value(rp, ci, B, cfg={"flags": strict})
        what = None
        for fl in (0, _lib.FLAG_BATCH_STREAM, _lib.FLAG_SEG_STREAM, _lib.FLAG_SHALLOW_UNROLL, _lib.FLAG_SC1_STORE | _lib.FLAG_SEG_STREAM,
                   _lib.FLAG_FORCE_IDX64, _lib.FLAG_SLAB_BLOCKED if (N * 4 >= 64 and K >= 256) else 0, _lib.FLAG_NO_XCD_REMAP | _lib.FLAG_NT_STORE):
            got = spmm.csr_spmm(rp, ci, val, B, cfg={"flags": fl | strict})
            nflags += 1
            if not torch.equal(got.view(torch.int32), ref.view(torch.int32)):
                what = "plain flags 0x%x" % fl
                break
        if what is None:
            for kernel in ("auto", "stream", "seg-stream"):
                plan = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=bool(rng.rand() < 0.8), kernel=kernel, flags=strict,
                                     task_entries=int(rng.choice([0, 8, 40, 200])))
                nplans += 1
                got = spmm.csr_spmm(rp, ci, val, B, plan=plan)
                if not torch.equal(got.view(torch.int32), ref.view(torch.int32)):
                    what = "plan kernel %s (%s)" % (kernel, plan.describe())
                    break
                gotu = spmm.csr_spmm_no_edge_value(rp, ci, B, plan=spmm.SpmmPlan(rp, ci, K, N, reorder=True, kernel=kernel, flags=strict))
                if not torch.equal(gotu.view(torch.int32), refu.view(torch.int32)):
                    what = "unweighted plan kernel %s" % kernel
                    break
        if what is not None:
            print("MISMATCH case %d shape %d M=%d K=%d nnz=%d N=%d: %s" % (case, shape, M, K, nnz, N, what), flush=True)
            return 1
        if case % 100 == 0:
            print("case %d ok (%.0f s, %d plain launches, %d plans)" % (case, time.time() - t0, nflags, nplans), flush=True)
    print("all %d cases equal bit for bit (%d plain launches, %d plans, %.0f s)" % (args.cases, nflags, nplans, time.time() - t0))
    return 0


if __name__ == "__main__":
    sys.exit(main())
