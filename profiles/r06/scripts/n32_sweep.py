#!/usr/bin/env python3
"""Headline graph at N = 32 (and 64): what the plan's knobs give — task size, row floor, unroll depth, kernel — before any new kernel is
written (round-5 review item 5)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402

from gespmm_amd import _lib, graphs, spmm  # noqa: E402
from kernel_ab import timeit  # noqa: E402

dev = torch.device("cuda")
g = graphs.synthetic_graph("com-amazon-sbm", seed=42, device=dev)
M, K, nnz, rp, ci = g["M"], g["K"], g["nnz"], g["rowptr"], g["colind"]
val = torch.rand(nnz, device=dev) - 0.5
for N in (32, 64):
    B = torch.rand(K, N, device=dev) - 0.5
    C = torch.empty((M, N), device=dev)
    alg = 4.0 * (M + 1) + 8.0 * nnz + 4.0 * (M + K) * N
    spmm.csr_spmm(rp, ci, val, B, out=C)
    ref = C.clone()
    p = spmm.SpmmPlan(rp, ci, K, N, values=val, expected_launches=1000000)
    t = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p), 50)
    print("N=%d AUTO steady-state plan %.1f us (%.3f) | %s" % (N, t, alg / t / 8e6, p.describe().split("|")[-1].strip()[:90]), flush=True)
    del p
    for kern in ("stream", "seg-stream", "staged"):
        for te in (0, 32, 64, 128, 256, 512):
            for rf in ((0, -1) if kern == "stream" else (0,)):
                for fl in ((0, 0x20000) if kern == "stream" else (0,)):
                    if kern == "staged" and te:
                        continue
                    try:
                        p = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True, kernel=kern, task_entries=te, row_floor=rf, flags=fl, expected_launches=1000000)
                    except Exception as ex:  # noqa: BLE001
                        print("  %s te=%d: %s" % (kern, te, str(ex)[:60]))
                        continue
                    C.zero_()
                    t = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p), 50)
                    ok = torch.equal(C.view(torch.int32), ref.view(torch.int32))
                    d = p.describe()
                    print("  N=%d %-10s task_entries=%-3d row_floor=%-2d flags=%#x: %.1f us (%.3f) tasks=%s%s" % (
                        N, kern, te, rf, fl, t, alg / t / 8e6, d.split("tasks=")[1].split(" ")[0] if "tasks=" in d else "-", "" if ok else " BITS-DIFFER"), flush=True)
                    del p
