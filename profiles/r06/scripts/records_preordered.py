#!/usr/bin/env python3
"""A matrix that ARRIVES clustered (the headline graph relabelled in its planted order): the plan keeps the storage order — and still takes
the padded-record kernel, judged by the modelled hits of the storage order.
    python profiles/r06/scripts/records_preordered.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402

from gespmm_amd import graphs, spmm  # noqa: E402
from kernel_ab import timeit  # noqa: E402

dev = torch.device("cuda")
g = graphs.synthetic_graph("com-amazon-sbm", seed=42, device=dev)
rp, ci = graphs.relabel_by_order(g["rowptr"], g["colind"], torch.argsort(g["truth"]))
M, K, nnz = g["M"], g["K"], g["nnz"]
val = torch.rand(nnz, device=dev) - 0.5
for N in (16, 32, 47, 64, 128):
    B = torch.rand(K, N, device=dev) - 0.5
    C = torch.empty((M, N), device=dev)
    alg = 4.0 * (M + 1) + 8.0 * nnz + 4.0 * (M + K) * N
    t_plain = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C), 30)
    ref = C.clone()
    row = "planted order N=%-3d plain %6.1f" % (N, t_plain)
    for label, kw in (("AUTO", {}), ("storage/stream", {"reorder": False, "kernel": "stream"}), ("storage/records", {"reorder": False, "kernel": "records"})):
        if label.endswith("records") and N > 64:
            continue
        p = spmm.SpmmPlan(rp, ci, K, N, values=val, expected_launches=1000000, **kw)
        C.zero_()
        t = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p), 30)
        ok = torch.equal(C.view(torch.int32), ref.view(torch.int32))
        d = p.describe()
        row += "  %s %6.1f us (%.3f) [%s %s]%s" % (label, t, alg / t / 8e6, d.split(" ")[0], d.split("|")[-1].strip().split(" ")[0][:22], "" if ok else " BITS-DIFFER")
        del p
    print(row, flush=True)
