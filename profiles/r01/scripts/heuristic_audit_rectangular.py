#!/usr/bin/env python3
"""Heuristic audit, rectangular shapes (bipartite / sampled blocks): AUTO against explicit configurations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gespmm_amd
from gespmm_amd import _lib as F, spmm

def time_fn(fn, iters, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

dev = torch.device("cuda:0")
torch.manual_seed(3)
cands = [("batch r2", dict(rows_per_wave=2, flags=F.FLAG_BATCH_STREAM | F.FLAG_NO_SLAB_BLOCKED)),
         ("batch r8", dict(rows_per_wave=8, flags=F.FLAG_BATCH_STREAM | F.FLAG_NO_SLAB_BLOCKED)),
         ("seg r2", dict(rows_per_wave=2, flags=F.FLAG_SEG_STREAM | F.FLAG_NO_SLAB_BLOCKED)),
         ("seg r8", dict(rows_per_wave=8, flags=F.FLAG_SEG_STREAM | F.FLAG_NO_SLAB_BLOCKED))]
bad = 0
for M, K in ((1000000, 1000), (1000000, 30000), (1000, 1000000), (30000, 1000000), (100000, 10000), (5000, 5000)):
    for deg in (5, 25, 100):
        nnz = M * deg
        rp = (torch.arange(M + 1, device=dev, dtype=torch.int64) * deg).to(torch.int32)
        ci = torch.randint(0, K, (nnz,), device=dev, dtype=torch.int32)
        val = torch.rand(nnz, device=dev)
        for N in (32, 128):
            B = torch.rand(K, N, device=dev); C = torch.empty(M, N, device=dev)
            it = 5 if nnz * N > 2e9 else 50
            auto = time_fn(lambda: spmm.csr_spmm(rp, ci, val, B, out=C), it)
            res = {name: time_fn(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, cfg=cfg), it) for name, cfg in cands}
            best = min(res, key=res.get)
            flag = "  <-- AUTO %.0f%% behind %s" % (100 * (auto / res[best] - 1), best) if auto > 1.10 * res[best] else ""
            bad += bool(flag)
            print("M=%8d K=%8d deg=%3d N=%3d auto %9.1f us | " % (M, K, deg, N, auto) + " | ".join("%s %.1f" % kv for kv in res.items()) + flag)
            sys.stdout.flush()
            del B, C
print("cases with AUTO > 10 %% behind: %d" % bad)
