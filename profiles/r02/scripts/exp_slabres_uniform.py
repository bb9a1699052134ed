#!/usr/bin/env python3
"""Upper bound of the slab-resident design: a hub-free graph of reddit's size whose rows all have the same degree
(232965 rows x 492 uniformly random columns), so static row ownership is perfectly balanced."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch

import gespmm_amd  # noqa: F401,E402
from gespmm_amd import spmm  # noqa: E402

dev = torch.device("cuda")
M, d, N = 232965, 492, 128
gen = torch.Generator(device=dev).manual_seed(5)
cols = torch.randint(0, M, (M, d), device=dev, generator=gen, dtype=torch.int32)
cols, _ = torch.sort(cols, dim=1)
rp = (torch.arange(M + 1, dtype=torch.int64, device=dev) * d).to(torch.int32)
ci = cols.reshape(-1).contiguous()
del cols
val = torch.rand(M * d, device=dev) - 0.5
B = torch.rand((M, N), device=dev)
C = torch.empty((M, N), device=dev)


def timeit(fn, iters=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for sr in (0, 6144, 8192, 12288):
    cfg = {"flags": 0x400, "slab_rows": sr}
    ms = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, cfg=cfg))
    print("slab-blocked slab_rows=%-6d %.3f ms" % (sr, ms), flush=True)
ref = C.clone()
for k in (0, 2, 4):
    os.environ["GESPMM_SLABRES_K"] = str(k)
    for sr in (2048, 3072, 4096, 6144, 8192):
        C.zero_()
        cfg = {"flags": 0x400 | 0x80000, "slab_rows": sr}
        ms = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, cfg=cfg))
        bad = (C.view(torch.int32) != ref.view(torch.int32)).any(dim=1).sum().item()
        print("resident k=%d slab_rows=%-6d %.3f ms   rows with different bits: %d" % (k, sr, ms, bad), flush=True)
