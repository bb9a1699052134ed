#!/usr/bin/env python3
"""L2-resident regime: B small enough for one XCD's L2, rows of a slab's worth of entries.
Which kernel structure gets closest to the pure-gather rate (microbench: 20-25 TB/s)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gespmm_amd
from gespmm_amd import _lib as F, graphs, spmm

def time_fn(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

dev = torch.device("cuda:0")
M = 232965
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
GRP = int(sys.argv[2]) if len(sys.argv) > 2 else 0
torch.manual_seed(0)
for K in (4096,):
    for deg in (8, 9, 17, 26, 32, 52):
        nnz = M * deg
        rp = (torch.arange(M + 1, device=dev, dtype=torch.int64) * deg).to(torch.int32)
        ci = torch.randint(0, K, (M, deg), device=dev, dtype=torch.int32).sort(dim=1).values.reshape(-1).contiguous()
        val = torch.rand(nnz, device=dev)
        B = torch.rand(K, N, device=dev); C = torch.empty(M, N, device=dev)
        line = "K=%5d (B %.1f MB) deg %2d gather %.2f GB:" % (K, K * N * 4 / 2**20, deg, nnz * N * 4 / 1e9)
        geo = dict(vec=4, strips=1, group=GRP) if GRP else {}
        for name, cfg in (("auto", None), ("seg", dict(flags=F.FLAG_SEG_STREAM | F.FLAG_NO_SLAB_BLOCKED, **geo)),
                          ("batch", dict(flags=F.FLAG_BATCH_STREAM | F.FLAG_NO_SLAB_BLOCKED, **geo)),
                          ("batch-u4", dict(flags=F.FLAG_BATCH_STREAM | F.FLAG_NO_SLAB_BLOCKED | F.FLAG_SHALLOW_UNROLL, **geo)),
                          ("slab1", dict(flags=F.FLAG_SLAB_BLOCKED, slab_rows=K, **geo)),
                          ("slab2", dict(flags=F.FLAG_SLAB_BLOCKED, slab_rows=K // 2, **geo))):
            us = time_fn(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, cfg=cfg))
            line += " %s %.0f us (%.1f TB/s) |" % (name, us, nnz * N * 4 / us / 1e6)
        print(line); sys.stdout.flush()
