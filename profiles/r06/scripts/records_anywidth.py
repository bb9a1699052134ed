#!/usr/bin/env python3
"""Padded-record kernel at widths that are not multiples of 4 (41 / 47 classes, 7, 20, 30 ...): 4-byte-aligned vector accesses, the lane at
the row's end takes the row's last four columns. Against the plain call and the AUTO plan (streaming kernels, one float per lane at odd widths).
    python profiles/r06/scripts/records_anywidth.py [graph ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402

from gespmm_amd import graphs, spmm  # noqa: E402
from kernel_ab import timeit  # noqa: E402

dev = torch.device("cuda")
for name in sys.argv[1:] or ["com-amazon-sbm", "products-sbm/4"]:
    g = graphs.synthetic_graph(name[:-2], seed=42, device=dev, scale=0.25) if name.endswith("/4") else graphs.synthetic_graph(name, seed=42, device=dev)
    M, K, nnz, rp, ci = g["M"], g["K"], g["nnz"], g["rowptr"], g["colind"]
    val = torch.rand(nnz, device=dev) - 0.5
    for N in (7, 10, 15, 16, 20, 30, 32, 41, 47, 48, 50, 62, 64):
        B = torch.rand(K, N, device=dev) - 0.5
        C = torch.empty((M, N), device=dev)
        alg = 4.0 * (M + 1) + 8.0 * nnz + 4.0 * (M + K) * N
        iters = 30 if nnz < 2e7 else 8
        t_plain = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C), iters)
        ref = C.clone()
        row = "%s N=%-3d plain %7.1f" % (name, N, t_plain)
        for label, kw in (("AUTO", {}), ("records", {"reorder": True, "kernel": "records"})):
            p = spmm.SpmmPlan(rp, ci, K, N, values=val, expected_launches=1000000, **kw)
            C.zero_()
            t = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p), iters)
            ok = torch.equal(C.view(torch.int32), ref.view(torch.int32))
            row += "  %s %7.1f us (%.3f) [%s]%s" % (label, t, alg / t / 8e6, p.describe().split("|")[-1].strip().split(" ")[0][:24], "" if ok else " BITS-DIFFER")
            del p
        print(row, flush=True)
