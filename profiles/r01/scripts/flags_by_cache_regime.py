#!/usr/bin/env python3
"""Controlled experiments on the com-Amazon-shaped rows: flags x cache regime."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gespmm_amd
from gespmm_amd import _lib, spmm

def time_fn(fn, iters=30, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

dev = torch.device("cuda:0")
M, N, deg = 334863, 128, 5.53
nnz = int(M * deg)
gen = torch.Generator(device=dev); gen.manual_seed(0)
rows = torch.sort(torch.randint(0, M, (nnz,), generator=gen, device=dev))[0]
rowptr = torch.zeros(M + 1, dtype=torch.int64, device=dev)
rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=M), 0)
rowptr = rowptr.to(torch.int32)
# regular-degree variant: every row exactly 8 entries (no tails, no empty rows)
rowptr8 = (torch.arange(M + 1, device=dev) * 8).to(torch.int32)
val = torch.rand(8 * M, device=dev) - 0.5
C = torch.empty((M, N), device=dev)
F = _lib
cfgs = [("seg g4", dict(rows_per_wave=4, flags=0)),
        ("seg g8", dict(rows_per_wave=8, flags=0)),
        ("seg g16", dict(rows_per_wave=16, flags=0)),
        ("seg g4 nt", dict(rows_per_wave=4, flags=F.FLAG_NT_STORE)),
        ("seg g4 u4", dict(rows_per_wave=4, flags=F.FLAG_SHALLOW_UNROLL)),
        ("bs r8", dict(rows_per_wave=8, flags=F.FLAG_BATCH_STREAM))]
for K in (2048, 334863, 1 << 22):
    B = torch.rand((K, N), device=dev)
    for name, rp, z in (("deg~5.5", rowptr, nnz), ("deg=8 exact", rowptr8, 8 * M)):
        colind = torch.randint(0, K, (z,), generator=gen, device=dev, dtype=torch.int32)
        for valued in (True, False):
            line = "K=%8d %-11s %-10s:" % (K, name, "valued" if valued else "unweighted")
            for label, cfg in cfgs:
                c = dict(vec=4, strips=1, group=32); c.update(cfg)
                if valued:
                    fn = lambda: spmm.csr_spmm(rp, colind, val[:z], B, variant=3, cfg=c, out=C)
                else:
                    fn = lambda: spmm.csr_spmm_no_edge_value(rp, colind, B, variant=3, cfg=c, out=C)
                line += " | %s %.1f" % (label, time_fn(fn))
            print(line); sys.stdout.flush()
    del B
