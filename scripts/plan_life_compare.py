#!/usr/bin/env python3
"""Launch time through a default-life plan (200 expected launches: three levels, three sweeps, 1 024 model samples) against a
steady-state plan (10^6: six levels, five sweeps, 4 096 samples) — and what each analysis cost.
    python scripts/plan_life_compare.py --graphs com-amazon-sbm ... --widths 128 256 512"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402

from gespmm_amd import spmm  # noqa: E402
import kernel_ab  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--graphs", nargs="+", required=True)
ap.add_argument("--widths", nargs="*", type=int, default=[128, 256, 512])
ap.add_argument("--lives", nargs="*", type=int, default=[200, 1000000])
args = ap.parse_args()
for name in args.graphs:
    g = kernel_ab.load(name, 1.0)
    M, K, nnz, rp, ci = g["M"], g["K"], g["nnz"], g["rowptr"], g["colind"]
    val = torch.rand(nnz, device="cuda") - 0.5
    iters = 30 if nnz < 8e6 else (10 if nnz < 5e7 else 4)
    for N in args.widths:
        B = torch.rand(K, N, device="cuda") - 0.5
        C = torch.empty((M, N), device="cuda")
        out = []
        for life in args.lives:
            spmm.SpmmPlan(rp, ci, K, N, values=val, expected_launches=life, reorder=True)  # warm (arena, code objects)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            p = spmm.SpmmPlan(rp, ci, K, N, values=val, expected_launches=life, reorder=True)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
            t = kernel_ab.timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p), iters)
            d = p.describe()
            out.append("life %-7d plan %6.2f ms launch %8.1f us (%s)" % (life, ms, t, d.split("kernel=")[1].split(" ")[0] if "kernel=" in d else "?"))
        print("%-16s N=%-4d %s" % (name, N, " | ".join(out)), flush=True)
