#!/usr/bin/env python3
"""Narrow/medium N on high-degree graphs: batch-stream vs segmented-stream vs naive."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gespmm_amd
from gespmm_amd import _lib as F, graphs, spmm

def time_fn(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

dev = torch.device("cuda:0")
for name in ("products-like", "reddit-like", "cit-hepth-like", "com-amazon-like"):
    gg = graphs.synthetic_graph(name, device=dev)
    v = torch.rand(gg["nnz"], device=dev)
    print("== %s avg deg %.1f" % (name, gg["nnz"] / gg["M"]))
    for N in (2, 4, 8, 16, 32, 64):
        B = torch.rand(gg["K"], N, device=dev); C = torch.empty(gg["M"], N, device=dev)
        line = "  N=%2d:" % N
        cands = [("auto", -1, None), ("v0", 0, None),
                 ("bs", -1, dict(flags=F.FLAG_BATCH_STREAM | F.FLAG_STRICT_ORDER)),
                 ("seg g1", -1, dict(rows_per_wave=1, flags=F.FLAG_SEG_STREAM | F.FLAG_STRICT_ORDER)),
                 ("seg g4", -1, dict(rows_per_wave=4, flags=F.FLAG_SEG_STREAM | F.FLAG_STRICT_ORDER)),
                 ("seg g4 u4", -1, dict(rows_per_wave=4, flags=F.FLAG_SEG_STREAM | F.FLAG_STRICT_ORDER | F.FLAG_SHALLOW_UNROLL)),
                 ("v1 bs", 1, dict(flags=F.FLAG_BATCH_STREAM | F.FLAG_STRICT_ORDER)),
                 ("v1 seg", 1, dict(rows_per_wave=2, flags=F.FLAG_SEG_STREAM | F.FLAG_STRICT_ORDER)),
                 ("v2 seg", 2, dict(rows_per_wave=2, flags=F.FLAG_SEG_STREAM | F.FLAG_STRICT_ORDER))]
        for label, variant, cfg in cands:
            us = time_fn(lambda: spmm.csr_spmm(gg["rowptr"], gg["colind"], v, B, variant=variant, cfg=cfg, out=C))
            line += " | %s %.0f" % (label, us)
        print(line); sys.stdout.flush()
        del B, C
    del gg, v
