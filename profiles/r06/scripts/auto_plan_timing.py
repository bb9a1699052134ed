#!/usr/bin/env python3
"""Round 6: the stateless entry points with gespmm_set_auto_plan on — per-call time (event pairs around groups of calls; the fingerprint's
synchronisation is inside) on the headline graph, N = 128: gespmm_csr_spmm_f32 (valued), gespmm_dgl_csrmm_sum_f32, against the same calls
with the switch off and against the plan held by the caller."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from gespmm_amd import _lib, graphs, spmm  # noqa: E402

P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
names = sys.argv[1:] or ["com-amazon-sbm", "com-amazon-like"]
for name in names:
    g = graphs.synthetic_graph(name, seed=42, device="cuda")
    rp, ci, M, K, nnz = g["rowptr"], g["colind"], g["M"], g["K"], g["nnz"]
    N = 128
    val = torch.rand(nnz, device="cuda") - 0.5
    B = torch.rand(K, N, device="cuda") - 0.5
    C = torch.empty(M, N, device="cuda")
    _lib.init(M, nnz)

    def wall(fn, n=200):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    calls = {
        "csr_spmm_f32 valued": lambda: _lib.lib.gespmm_csr_spmm_f32(P(rp), P(ci), P(val), P(B), P(C), M, K, N, nnz, -1, None),
        "csr_spmm_f32 unweighted": lambda: _lib.lib.gespmm_csr_spmm_f32(P(rp), P(ci), None, P(B), P(C), M, K, N, nnz, -1, None),
        "dgl_csrmm_sum": lambda: _lib.lib.gespmm_dgl_csrmm_sum_f32(M, N, P(rp), P(ci), P(B), P(C), None),
        "dgl_csrmm_max": lambda: _lib.lib.gespmm_dgl_csrmm_max_f32(M, N, P(rp), P(ci), P(B), P(C), None),
    }
    plan = spmm.SpmmPlan(rp, ci, K, N, values=val)
    held = wall(lambda: _lib.lib.gespmm_plan_spmm_f32(plan._handle, P(B), P(C), N, None))
    print("%s N=%d | plan held by the caller %.1f us per call (wall clock, 200 calls)" % (name, N, held), flush=True)
    del plan
    for label, fn in calls.items():
        _lib.set_auto_plan(0)
        off = wall(fn)
        _lib.set_auto_plan(3)
        s0 = _lib.auto_plan_stats()
        first = []
        for i in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            first.append((time.perf_counter() - t0) * 1e6)
        on = wall(fn)
        s1 = _lib.auto_plan_stats()
        print("  %-24s switch off %.1f us | on: calls 1-4 %s us, then %.1f us per call (planned %d of them %d without a synchronisation, fingerprints read back %d, plans %d)" % (
            label, off, " ".join("%.0f" % t for t in first), on, s1["calls_planned"] - s0["calls_planned"], s1["calls_async"] - s0["calls_async"], s1["fingerprints"] - s0["fingerprints"],
            s1["plans_created"] - s0["plans_created"]), flush=True)
    _lib.set_auto_plan(0)
    del g, rp, ci, val, B, C
    torch.cuda.empty_cache()
