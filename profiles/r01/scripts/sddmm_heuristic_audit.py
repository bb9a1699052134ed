#!/usr/bin/env python3
"""SDDMM heuristic audit: CSR form (may take the cache-blocked kernel) against the COO form (always
streaming) on a grid of uniform-degree random patterns."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gespmm_amd
from gespmm_amd import sddmm

def time_fn(fn, iters, warm=2):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

dev = torch.device("cuda:0")
torch.manual_seed(2)
for M in (20000, 100000, 200000, 1000000):
    for deg in (64, 100, 150, 300, 600):
        nnz = M * deg
        if nnz > 1.3e8: continue
        rp = (torch.arange(M + 1, device=dev, dtype=torch.int64) * deg).to(torch.int32)
        ci = torch.randint(0, M, (M, deg), device=dev, dtype=torch.int32).sort(dim=1).values.reshape(-1).contiguous()
        rows = torch.arange(M, device=dev, dtype=torch.int32).repeat_interleave(deg)
        for N in (64, 128, 256):
            D1 = torch.rand(M, N, device=dev); D2 = torch.rand(M, N, device=dev)
            it = 3 if nnz * N > 2e9 else 20
            a = time_fn(lambda: sddmm.csr_sddmm(rp, ci, D1, D2), it)
            b = time_fn(lambda: sddmm.coo_sddmm(rows, ci, D1, D2), it)
            nslab = -(-M // ((6 << 20) // (N * 4)))
            print("M=%8d deg=%3d N=%3d slabs(6MB)=%3d  csr %9.1f us  coo %9.1f us  csr/coo %.2f%s" % (
                M, deg, N, nslab, a, b, a / b, "  <-- CSR slower" if a > 1.1 * b else ""))
            sys.stdout.flush()
            del D1, D2
