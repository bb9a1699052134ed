#!/usr/bin/env python3
"""Known-byte-count launches of the plain batch-stream kernel, to calibrate FETCH_SIZE for THIS access pattern
(512-byte B rows gathered by half-wavefronts): every row has exactly `deg` non-zeros whose columns are `deg` independent
permutations of 0..M-1, so the kernel reads every B row exactly `deg` times (no more, no fewer); `--identity` makes the
columns i, i+1.. (a streaming read).  Run under rocprofv3 --pmc FETCH_SIZE etc. (scripts/gpu_pmc.sh).

    python profiles/r02/scripts/fetch_calibration.py --deg 1 [--identity] [--iters 50]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
ap = argparse.ArgumentParser()
ap.add_argument("--deg", type=int, default=1)
ap.add_argument("--identity", action="store_true")
ap.add_argument("--degs", default="", help="comma list: row i has degs[i % len] non-zeros (columns uniformly random)")
ap.add_argument("--graph", default="", help="a bench graph (gespmm_amd.graphs) instead of the synthetic pattern")
ap.add_argument("--k", type=int, default=0, help="with --degs: columns drawn from 0..k-1 (B has k rows; small k = all gathers hit L2)")
ap.add_argument("--flags", type=lambda x: int(x, 0), default=0)
ap.add_argument("--rpw", type=int, default=0, help="rows per wavefront / lane group (launch cfg)")
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--m", type=int, default=334863)
ap.add_argument("--n", type=int, default=128)
args = ap.parse_args()
import torch

import gespmm_amd  # noqa: F401,E402
from gespmm_amd import spmm  # noqa: E402

dev = torch.device("cuda")
M, N, d = args.m, args.n, args.deg
g = torch.Generator(device="cpu").manual_seed(7)
gathers = None
if args.graph:
    from gespmm_amd import graphs
    gg = graphs.synthetic_graph(args.graph, seed=42, device=dev)
    M, rp, ci = gg["M"], gg["rowptr"], gg["colind"]
    gathers = gg["nnz"]
    K = gg["K"]
elif args.degs:
    dl = torch.tensor([int(x) for x in args.degs.split(",")])
    degs = dl[torch.arange(M) % len(dl)]
    rp64 = torch.zeros(M + 1, dtype=torch.int64)
    rp64[1:] = torch.cumsum(degs, 0)
    gathers = int(rp64[-1])
    rows = torch.repeat_interleave(torch.arange(M), degs)
    K = args.k if args.k > 0 else M
    colsr = torch.randint(0, K, (gathers,), generator=g)
    order = torch.argsort(rows * K + colsr)
    rp, ci = rp64.to(torch.int32).to(dev), colsr[order].to(torch.int32).to(dev)
if gathers is not None:
    val = torch.rand(gathers, device=dev) - 0.5
    B = torch.rand((K, N), device=dev)
    C = torch.empty((M, N), device=dev)
    d = gathers / M
elif args.identity:
    cols = (torch.arange(M).unsqueeze(1) + torch.arange(d).unsqueeze(0)) % M
else:
    cols = torch.stack([torch.randperm(M, generator=g) for _ in range(d)], dim=1)
if gathers is None:
    cols, _ = torch.sort(cols, dim=1)
    rp = (torch.arange(M + 1, dtype=torch.int64) * d).to(torch.int32).to(dev)
    ci = cols.reshape(-1).to(torch.int32).to(dev)
    val = torch.rand(M * d, device=dev) - 0.5
    B = torch.rand((M, N), device=dev)
    C = torch.empty((M, N), device=dev)
    gathers = M * d
for _ in range(5):
    spmm.csr_spmm(rp, ci, val, B, out=C, cfg=({"flags": args.flags, "rows_per_wave": args.rpw} if (args.flags or args.rpw) else None))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(args.iters):
    spmm.csr_spmm(rp, ci, val, B, out=C, cfg=({"flags": args.flags, "rows_per_wave": args.rpw} if (args.flags or args.rpw) else None))
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / args.iters * 1e3
row_bytes = N * 4
print("%s: %.1f us | %d gathers of %d B = %.1f MB = %d lines of 128 B, meta %.1f MB, C %.1f MB" % (
    " ".join(sys.argv[1:]), us, gathers, row_bytes, gathers * row_bytes / 1e6, gathers * row_bytes // 128,
    (8 * gathers + 4 * M) / 1e6, M * row_bytes / 1e6))
