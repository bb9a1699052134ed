#!/usr/bin/env python3
"""Secondary ops and configs: SDDMM, csr2csc, narrow N (variant 5), products-like N sweep."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gespmm_amd
from gespmm_amd import _lib as F, graphs, spmm, sddmm

def time_fn(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

dev = torch.device("cuda:0")
g = graphs.synthetic_graph("com-amazon-like", device=dev)
rp, ci, M, nnz = g["rowptr"], g["colind"], g["M"], g["nnz"]
rows = torch.repeat_interleave(torch.arange(M, device=dev), (rp[1:] - rp[:-1]).long()).to(torch.int32)
val = torch.rand(nnz, device=dev) - 0.5
print("== SDDMM com-amazon-like (nnz=%d): bytes gathered = 2*N*4 per edge" % nnz)
for N in (3, 32, 100, 128, 256):
    D1 = torch.rand(M, N, device=dev); D2 = torch.rand(M, N, device=dev)
    us_coo = time_fn(lambda: sddmm.coo_sddmm(rows, ci, D1, D2))
    us_csr = time_fn(lambda: sddmm.csr_sddmm(rp, ci, D1, D2))
    us_ref = time_fn(lambda: (D1[rows.long()] * D2[ci.long()]).sum(1))
    print("  N=%3d coo %.1f us (%.2f TB/s gathered)  csr %.1f us   torch gather-mul-sum %.1f us" %
          (N, us_coo, 8.0 * N * nnz / us_coo / 1e6, us_csr, us_ref))
print("== csr2csc com-amazon-like")
colptr = torch.empty(M + 1, dtype=torch.int32, device=dev); rowind = torch.empty(nnz, dtype=torch.int32, device=dev)
us = time_fn(lambda: spmm.csr2csc(rp, ci, colptr, rowind, val))
us_t = time_fn(lambda: graphs.transpose_csr(rp, ci, val=val))
print("  csr2csc %.1f us ; torch argsort-based transpose %.1f us" % (us, us_t))
print("== narrow N on pubmed-selfloop-like / com-amazon-like: auto vs parallel-reduction")
for name in ("pubmed-selfloop-like", "com-amazon-like", "reddit-like"):
    gg = graphs.synthetic_graph(name, device=dev)
    v = torch.rand(gg["nnz"], device=dev)
    for N in (1, 2, 3, 4, 8, 16):
        B = torch.rand(gg["K"], N, device=dev)
        line = "  %-22s N=%2d:" % (name, N)
        for label, variant, cfg in (("auto", -1, None), ("v0", 0, None), ("v5", 5, None), ("v5 W8", 5, dict(group=8)), ("v5 W64", 5, dict(group=64))):
            us = time_fn(lambda: spmm.csr_spmm(gg["rowptr"], gg["colind"], v, B, variant=variant, cfg=cfg))
            line += " | %s %.1f us" % (label, us)
        print(line); sys.stdout.flush()
    del gg
print("== products-like sweep (BASELINE config 3): N in 16..512, auto vs each variant")
gg = graphs.synthetic_graph("products-like", device=dev)
v = torch.rand(gg["nnz"], device=dev)
for N in (16, 32, 64, 128, 256, 512):
    B = torch.rand(gg["K"], N, device=dev); C = torch.empty(gg["M"], N, device=dev)
    ab = 4 * (gg["M"] + 1) + 8 * gg["nnz"] + 4 * gg["K"] * N + 4 * gg["M"] * N
    line = "  N=%3d (alg %.0f MB):" % (N, ab / 1e6)
    for variant in (-1, 1, 2, 3, 4):
        us = time_fn(lambda: spmm.csr_spmm(gg["rowptr"], gg["colind"], v, B, variant=variant, out=C), iters=4)
        line += " | v%d %.0f us %.2f TF frac %.3f" % (variant, us, 2.0 * gg["nnz"] * N / us / 1e6, ab / us / 1e3 / 8000)
    print(line); sys.stdout.flush()
    del B, C
