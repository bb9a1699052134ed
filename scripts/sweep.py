#!/usr/bin/env python3
"""Geometry / variant sweep of the SpMM kernels on one GPU (tuning aid, not a test).

    python scripts/sweep.py [--graphs a,b] [--ncols 32,128,512] [--out gpurun_out/sweep.json] [--quick]

For every (graph, N, valued) it times each variant and a grid of explicit launch
geometries (vec x strips x group x flags) with HIP events on the launch stream,
interleaved in ONE process (cdna_hip_programming.md §5.4 rule 24), and reports
microseconds, GFLOP/s and algorithmic GB/s. A device copy of the same footprint is
timed alongside as the achievable-bandwidth yardstick."""
import argparse
import itertools
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import gespmm_amd  # noqa: E402,F401
from gespmm_amd import _lib, graphs, spmm  # noqa: E402


def time_fn(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", default="com-amazon-like,com-amazon-like@0.9")
    ap.add_argument("--ncols", default="32,128,512")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sweep.json"))
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    results = []
    for gname in args.graphs.split(","):
        name, _, loc = gname.partition("@")
        g = graphs.synthetic_graph(name, seed=42, device=dev, locality=float(loc) if loc else 0.0)
        M, K, nnz = g["M"], g["K"], g["nnz"]
        rp, ci = g["rowptr"], g["colind"]
        val = torch.rand(nnz, device=dev) - 0.5
        iters = args.iters if nnz < 2e7 else max(args.iters // 6, 3)
        for N in [int(x) for x in args.ncols.split(",")]:
            B = (torch.randint(0, 100, (K, N), device=dev, dtype=torch.int32) - 50).float() / 100
            C = torch.empty((M, N), device=dev)
            copy_us = time_fn(lambda: C.copy_(B[:M]), iters)
            for valued in (True, False):
                ab = 4 * (M + 1) + 4 * nnz + (4 * nnz if valued else 0) + 4 * K * N + 4 * M * N
                cands = [("v%d" % v, v, None) for v in range(6)]
                vecs = [v for v in (1, 2, 4) if N % v == 0]
                vmax = max(vecs)
                geos = []
                for strips in (1, 2):
                    if strips == 2 and vmax != 4:
                        continue
                    per = vmax * strips
                    w = 4
                    while w < 64 and w * per < N:
                        w *= 2
                    geos.append((vmax, strips, w))
                    if w * per >= 2 * N or w == 64:
                        pass
                    if w > 4 and not args.quick:
                        geos.append((vmax, strips, w // 2))  # two column tiles per row
                if vmax == 4 and N % 2 == 0 and not args.quick:
                    w = 4
                    while w < 64 and w * 2 < N:
                        w *= 2
                    geos.append((2, 1, w))
                for vec, strips, grp in geos:
                    G = 64 // grp
                    base = {"vec": vec, "strips": strips, "group": grp}
                    for rpw in sorted({G, 8, 16}):  # batch-stream kernel (rows per wavefront)
                        if rpw < G or rpw > 32:
                            continue
                        cands.append(("V%d S%d W%d bs r%d" % (vec, strips, grp, rpw), 1,
                                      dict(base, rows_per_wave=rpw, flags=_lib.FLAG_BATCH_STREAM)))
                    for rpg in (1, 2, 4, 8, 16, 32):  # segmented-stream kernel (rows per lane group)
                        for fl, tag in ((0, ""), (_lib.FLAG_SHALLOW_UNROLL, " u4")):
                            if strips == 2 and fl:
                                continue
                            cands.append(("V%d S%d W%d seg g%d%s" % (vec, strips, grp, rpg, tag), 1,
                                          dict(base, rows_per_wave=rpg, flags=fl)))
                    if not args.quick:
                        for fl, tag in ((_lib.FLAG_NT_STORE, " nt"), (_lib.FLAG_NO_XCD_REMAP, " noxcd"),
                                        (_lib.FLAG_FORCE_IDX64, " i64")):
                            cands.append(("V%d S%d W%d seg g8%s" % (vec, strips, grp, tag), 1,
                                          dict(base, rows_per_wave=8, flags=fl)))
                best = {}
                for _ in range(args.rounds):
                    for label, variant, cfg in cands:
                        def fn():
                            if valued:
                                spmm.csr_spmm(rp, ci, val, B, variant=variant, cfg=cfg, out=C)
                            else:
                                spmm.csr_spmm_no_edge_value(rp, ci, B, variant=variant, cfg=cfg, out=C)
                        us = time_fn(fn, iters, warm=2)
                        best.setdefault(label, []).append(us)
                rows = []
                for label, v in best.items():
                    v = sorted(v)
                    med = v[len(v) // 2]
                    rows.append({"cfg": label, "us_med": med, "us_min": v[0], "gflops": 2.0 * nnz * N / med / 1e3,
                                 "GBs": ab / med / 1e3, "frac": ab / med / 1e3 / 8000.0})
                rows.sort(key=lambda r: r["us_med"])
                print("== %s N=%d %s  M=%d nnz=%d  alg=%.1f MB  copy(same C+B bytes)=%.1f us (%.0f GB/s)" %
                      (gname, N, "valued" if valued else "unweighted", M, nnz, ab / 1e6, copy_us,
                       8.0 * M * N / copy_us / 1e3))
                for r in rows[:14] + [r for r in rows[14:] if r["cfg"].startswith("v")]:
                    print("   %-22s %9.1f us (min %9.1f)  %9.1f GFLOP/s  %7.1f GB/s  frac %.3f" %
                          (r["cfg"], r["us_med"], r["us_min"], r["gflops"], r["GBs"], r["frac"]))
                sys.stdout.flush()
                results.append({"graph": gname, "N": N, "valued": valued, "M": M, "nnz": nnz, "alg_bytes": ab,
                                "copy_us": copy_us, "rows": rows})
            del B, C
        del g, rp, ci, val
        torch.cuda.empty_cache()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(results, f)


if __name__ == "__main__":
    main()
