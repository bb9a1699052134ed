#!/usr/bin/env python3
"""One (graph, width, plan kernel) a few launches — the command scripts/gpu_pmc.sh profiles.
    python scripts/kernel_pmc_case.py <graph> <N> <stream|seg-stream|staged|auto|plain> [launches]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402

from gespmm_amd import spmm  # noqa: E402
import kernel_ab  # noqa: E402

name, N, kern = sys.argv[1], int(sys.argv[2]), sys.argv[3]
launches = int(sys.argv[4]) if len(sys.argv) > 4 else 5
g = kernel_ab.load(name, 1.0)
M, K, nnz, rp, ci = g["M"], g["K"], g["nnz"], g["rowptr"], g["colind"]
val = torch.rand(nnz, device="cuda") - 0.5
B = torch.rand(K, N, device="cuda") - 0.5
C = torch.empty((M, N), device="cuda")
plan = None
if kern != "plain":
    plan = spmm.SpmmPlan(rp, ci, K, N, values=val, **({} if kern == "auto" else {"reorder": True, "kernel": kern}))
    print(plan.describe())
for _ in range(launches):
    spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan)
torch.cuda.synchronize()
print("alg bytes %d" % (4 * (M + 1) + 8 * nnz + 4 * (M + K) * N))
