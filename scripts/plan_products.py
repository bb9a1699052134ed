#!/usr/bin/env python3
"""Clustered plans on the products-shaped graphs (123.7 M non-zeros): analysis time and kernel time, N sweep."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import gespmm_amd  # noqa: F401,E402
from gespmm_amd import graphs, spmm  # noqa: E402

dev = torch.device("cuda")


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name in sys.argv[1:] or ["products-sbm", "products-like"]:
    t0 = time.time()
    g = graphs.synthetic_graph(name, seed=42, device=dev)
    torch.cuda.synchronize()
    print(name, "generated in %.1f s" % (time.time() - t0), "M", g["M"], "nnz", g["nnz"], flush=True)
    M, K, nnz = g["M"], g["K"], g["nnz"]
    rp, ci = g["rowptr"], g["colind"]
    val = torch.rand(nnz, device=dev) - 0.5
    plans = {}
    for N in (128, 32, 512):
        B = torch.rand(K, N, device=dev) - 0.5
        C = torch.empty((M, N), device=dev)
        ab = 4 * (M + 1) + 8 * nnz + 4 * K * N + 4 * M * N
        ms = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C))
        ref = C.clone()
        print("%s N=%d plain  %8.3f ms  %6.2f TFLOP/s  frac %.3f" % (name, N, ms, 2.0 * nnz * N / ms / 1e9, ab / ms / 8e9), flush=True)
        t0 = time.time()
        plan = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True)
        dt = time.time() - t0
        ms = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan))
        ok = torch.equal(C.view(torch.int32), ref.view(torch.int32))
        print("%s N=%d plan   %8.3f ms  %6.2f TFLOP/s  frac %.3f  bits_equal=%s  analysis %.1f s | %s" %
              (name, N, ms, 2.0 * nnz * N / ms / 1e9, ab / ms / 8e9, ok, dt, plan.describe()), flush=True)
        del plan, B, C, ref
        torch.cuda.empty_cache()
