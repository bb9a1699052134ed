#!/usr/bin/env python3
"""RMAT shard x N: column tiles of 32 / 64 / 128 columns bound to XCDs (tile = workgroup id mod ntile, ids go to XCDs round-robin), so
an XCD's 4 MiB L2 holds only ITS slice of every B row: 8 / 4 / 2 times the rows. Costs reading A once per tile.
    python profiles/r06/scripts/rmat_xcd_tiles.py [scale ...]"""
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


widths = [int(x) for x in os.environ.get("WIDTHS", "256").split(",")]
for scale in [int(s) for s in sys.argv[1:]] or [22, 24]:
    g = graphs.rmat_shard(scale, device=dev)
    M, K, nnz, rp, ci = g["M"], g["K"], g["nnz"], g["rowptr"], g["colind"]
    val = torch.rand(nnz, device=dev) - 0.5
    for N in widths:
        B = torch.rand(K, N, device=dev) - 0.5
        C = torch.empty((M, N), device=dev)
        alg = 4.0 * (M + 1) + 8.0 * nnz + 4.0 * (M + K) * N
        iters = 6 if scale <= 22 else 3
        spmm.csr_spmm(rp, ci, val, B, out=C)
        ref = C.clone()
        tol = 1e-4 * float(ref.abs().max())
        print("rmat-%d N=%d M=%d nnz=%d" % (scale, N, M, nnz), flush=True)
        X = _lib.FLAG_NO_XCD_REMAP
        cases = [("default", {})]
        for grp in (8, 16, 32):
            if grp * 4 >= N:
                continue
            for rpw in (0, 8, 32):
                cases.append(("group=%d bound rpw=%d" % (grp, rpw), {"vec": 4, "strips": 1, "group": grp, "rows_per_wave": rpw, "flags": X}))
            cases.append(("group=%d bound shallow" % grp, {"vec": 4, "strips": 1, "group": grp, "flags": X | _lib.FLAG_SHALLOW_UNROLL}))
            cases.append(("group=%d NOT bound" % grp, {"vec": 4, "strips": 1, "group": grp}))
        for name, cfg in cases:
            C.zero_()
            try:
                t = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, cfg=cfg or None, out=C), iters)
            except Exception as e:  # noqa: BLE001
                print("  %-28s %s" % (name, e), flush=True)
                continue
            d = (C - ref).abs().max().item()
            print("  %-28s %8.3f ms  frac %.3f  maxdiff %.2e%s" % (name, t, alg / (t * 1e-3) / 8e12, d, "" if d <= tol else "  !!"), flush=True)
        del B, C, ref
        torch.cuda.empty_cache()
    del g, rp, ci, val
    torch.cuda.empty_cache()
