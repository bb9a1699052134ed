#!/usr/bin/env python3
"""Time of gespmm_plan_create (AUTO, defaults) on named graphs: best / median of `reps` creations after the first (which loads the analysis
kernels), and what the plan decided.    python scripts/plan_ms.py [--widths 128] [--reps 7] graph ..."""
import argparse
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402

from gespmm_amd import spmm  # noqa: E402
import kernel_ab  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("graphs", nargs="+")
ap.add_argument("--widths", nargs="*", type=int, default=[128])
ap.add_argument("--reps", type=int, default=7)
ap.add_argument("--expected-launches", type=int, default=0)
args = ap.parse_args()
for name in args.graphs:
    g = kernel_ab.load(name, 1.0)
    val = torch.rand(g["nnz"], device="cuda") - 0.5
    for N in args.widths:
        ts = []
        for _ in range(args.reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            p = spmm.SpmmPlan(g["rowptr"], g["colind"], g["K"], N, values=val, expected_launches=args.expected_launches)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
            d = p.describe()
            del p
        print("%-16s N=%-4d first %.2f ms, then best %.2f median %.2f ms | %s" % (name, N, ts[0], min(ts[1:]), statistics.median(ts[1:]), d[:230]),
              flush=True)
