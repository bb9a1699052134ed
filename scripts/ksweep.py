#!/usr/bin/env python3
"""Cache-regime sweep: the SpMM kernel on com-Amazon-shaped rows (M = 334 863, ~5.5
nnz/row, N = 128) whose column indices are uniform in [0, K). K sets the footprint
of B: a few MB (lives in every XCD's 4 MiB L2), tens of MB (Infinity Cache), GBs
(HBM). Shows which level of the hierarchy binds the gather. Also times plain
streaming reads of the same footprints for reference."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import gespmm_amd  # noqa: E402,F401
from gespmm_amd import spmm  # noqa: E402


def time_fn(fn, iters=30, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    M, N, deg = 334863, 128, 5.53
    nnz = int(M * deg)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)
    rows = torch.sort(torch.randint(0, M, (nnz,), generator=gen, device=dev))[0]
    rowptr = torch.zeros(M + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=M), 0)
    rowptr = rowptr.to(torch.int32)
    val = torch.rand(nnz, device=dev) - 0.5
    C = torch.empty((M, N), device=dev)
    print("M=%d nnz=%d N=%d ; gather bytes/launch = %.0f MB, C = %.0f MB" % (M, nnz, N, 4.0 * nnz * N / 1e6, 4.0 * M * N / 1e6))
    for K in (2048, 8192, 32768, 131072, 334863, 1 << 20, 1 << 22):
        colind = torch.randint(0, K, (nnz,), generator=gen, device=dev, dtype=torch.int32)
        B = torch.rand((K, N), device=dev)
        us = time_fn(lambda: spmm.csr_spmm(rowptr, colind, val, B, out=C))
        seq = (torch.arange(nnz, device=dev) % K).to(torch.int32)  # consecutive B rows: streaming-like
        us_seq = time_fn(lambda: spmm.csr_spmm(rowptr, seq, val, B, out=C))
        total = 4.0 * nnz * N + 4.0 * M * N + 8.0 * nnz
        print("K=%8d  B=%8.1f MB : random cols %8.1f us (%.2f TB/s of gather+store)   sequential cols %8.1f us" %
              (K, 4.0 * K * N / 1e6, us, total / us / 1e6, us_seq))
        del B, colind
    for mb in (32, 128, 171, 343, 1024, 4096):
        x = torch.rand(mb * 250000, device=dev)
        us = time_fn(lambda: x.sum())
        y = torch.empty_like(x)
        us_c = time_fn(lambda: y.copy_(x))
        print("stream read %5d MB: %8.1f us = %.2f TB/s ;  copy: %8.1f us = %.2f TB/s (r+w)" %
              (mb, us, mb / us, us_c, 2.0 * mb / us_c))
        del x, y


if __name__ == "__main__":
    main()
