#!/usr/bin/env python3
"""rows-per-group of the segmented kernel across cache regimes (interleaved rounds)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gespmm_amd
from gespmm_amd import _lib as F, graphs, spmm

def time_fn(fn, iters=40, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

dev = torch.device("cuda:0")
cfgs = [("seg g1", dict(rows_per_wave=1)), ("seg g2", dict(rows_per_wave=2)), ("seg g3", dict(rows_per_wave=3)),
        ("seg g4", dict(rows_per_wave=4)), ("seg g8", dict(rows_per_wave=8)),
        ("seg g1 u4", dict(rows_per_wave=1, flags=F.FLAG_SHALLOW_UNROLL)), ("seg g2 u4", dict(rows_per_wave=2, flags=F.FLAG_SHALLOW_UNROLL)),
        ("bs r2", dict(rows_per_wave=2, flags=F.FLAG_BATCH_STREAM)), ("bs r4", dict(rows_per_wave=4, flags=F.FLAG_BATCH_STREAM)),
        ("bs r8", dict(rows_per_wave=8, flags=F.FLAG_BATCH_STREAM)), ("V2W64 seg g1", dict(vec=2, group=64, rows_per_wave=1)),
        ("V2W64 seg g2", dict(vec=2, group=64, rows_per_wave=2)), ("V2W64 bs r4", dict(vec=2, group=64, rows_per_wave=4, flags=F.FLAG_BATCH_STREAM))]
for name, loc in (("com-amazon-like", 0.0), ("com-amazon-like", 0.9), ("cit-hepth-like", 0.0), ("pubmed-selfloop-like", 0.0)):
    g = graphs.synthetic_graph(name, device=dev, locality=loc)
    val = torch.rand(g["nnz"], device=dev) - 0.5
    B = torch.rand(g["K"], 128, device=dev); C = torch.empty(g["M"], 128, device=dev)
    res = {c[0]: [] for c in cfgs}
    for _ in range(4):
        for label, cfg in cfgs:
            res[label].append(time_fn(lambda: spmm.csr_spmm(g["rowptr"], g["colind"], val, B, variant=3, cfg=cfg, out=C)))
    print("== %s locality %.1f N=128:" % (name, loc) + "".join(" | %s %.1f" % (k, sorted(v)[len(v) // 2]) for k, v in res.items()))
    sys.stdout.flush()
