#!/usr/bin/env python3
"""RMAT shard x N = 256 through the plain call's knobs (round 6, before the record-stream form): what the existing launch shapes give.
    python profiles/r06/scripts/rmat_knobs.py [scale ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from gespmm_amd import _lib, graphs, spmm  # noqa: E402

dev = torch.device("cuda")


def timeit(fn, iters):
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


N = 256
for scale in [int(s) for s in sys.argv[1:]] or [22, 24]:
    g = graphs.rmat_shard(scale, device=dev)
    M, K, nnz, rp, ci = g["M"], g["K"], g["nnz"], g["rowptr"], g["colind"]
    deg = (rp[1:] - rp[:-1]).long()
    print("rmat-%d M=%d nnz=%d empty_rows=%.3f rows>2048: %d holding %.3f of the entries, rows 33..2048: %d holding %.3f" % (
        scale, M, nnz, float((deg == 0).float().mean()), int((deg > 2048).sum()), float(deg[deg > 2048].sum()) / nnz,
        int(((deg > 32) & (deg <= 2048)).sum()), float(deg[(deg > 32) & (deg <= 2048)].sum()) / nnz), flush=True)
    val = torch.rand(nnz, device=dev) - 0.5
    B = torch.rand(K, N, device=dev) - 0.5
    C = torch.empty((M, N), device=dev)
    alg = 4.0 * (M + 1) + 8.0 * nnz + 4.0 * (M + K) * N
    iters = 6 if scale <= 22 else 3
    spmm.csr_spmm(rp, ci, val, B, out=C)
    ref = C.clone()
    cases = [("default", {}), ("shallow(U=4)", {"flags": _lib.FLAG_SHALLOW_UNROLL})]
    for rpw in (2, 4, 8, 16):
        cases.append(("rpw=%d" % rpw, {"rows_per_wave": rpw}))
        cases.append(("rpw=%d shallow" % rpw, {"rows_per_wave": rpw, "flags": _lib.FLAG_SHALLOW_UNROLL}))
    cases.append(("no xcd remap", {"flags": _lib.FLAG_NO_XCD_REMAP}))
    cases.append(("no xcd remap rpw=4", {"flags": _lib.FLAG_NO_XCD_REMAP, "rows_per_wave": 4}))
    cases.append(("no xcd remap shallow", {"flags": _lib.FLAG_NO_XCD_REMAP | _lib.FLAG_SHALLOW_UNROLL}))
    cases.append(("strict(no long-row pass)", {"flags": _lib.FLAG_STRICT_ORDER}))
    cases.append(("nt stores", {"flags": _lib.FLAG_NT_STORE}))
    cases.append(("sc1 stores", {"flags": _lib.FLAG_SC1_STORE}))
    for name, cfg in cases:
        C.zero_()
        t = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, cfg=cfg or None, out=C), iters)
        d = (C - ref).abs().max().item()
        print("  %-26s %8.3f ms  frac %.3f  maxdiff %.2e" % (name, t, alg / (t * 1e-3) / 8e12, d), flush=True)
    del g, rp, ci, val, B, C, ref
    torch.cuda.empty_cache()
