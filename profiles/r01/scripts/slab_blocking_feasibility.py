#!/usr/bin/env python3
"""Feasibility of column-slab cache blocking on reddit-like: run the existing kernel on
A restricted to one slab of B rows at a time (slab small enough to live in a 4 MiB L2)
and add up the times. No accumulation of C here — this only measures the gather side."""
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
g = graphs.synthetic_graph("reddit-like", device=dev)
M, K, nnz = g["M"], g["K"], g["nnz"]
rp, ci = g["rowptr"], g["colind"]
N = 128
B = torch.rand(K, N, device=dev); C = torch.empty(M, N, device=dev)
val = torch.rand(nnz, device=dev)
full = time_fn(lambda: spmm.csr_spmm(rp, ci, val, B, out=C))
print("full kernel: %.0f us" % full)
rows = torch.repeat_interleave(torch.arange(M, device=dev), (rp[1:] - rp[:-1]).long())
for slab_rows in (2048, 4096, 8192, 16384):
    nslab = (K + slab_rows - 1) // slab_rows
    subs = []
    for s in range(nslab):
        m = (ci >= s * slab_rows) & (ci < (s + 1) * slab_rows)
        sub_ci = ci[m].contiguous(); sub_v = val[m].contiguous()
        sub_rp = torch.zeros(M + 1, dtype=torch.int64, device=dev)
        sub_rp[1:] = torch.cumsum(torch.bincount(rows[m], minlength=M), 0)
        subs.append((sub_rp.to(torch.int32), sub_ci, sub_v))
    def run_all():
        for sub_rp, sub_ci, sub_v in subs:
            spmm.csr_spmm(sub_rp, sub_ci, sub_v, B, out=C)
    us = time_fn(run_all, iters=3, warm=1)
    print("slab %6d rows (%.1f MB of B), %3d launches: total %.0f us (%.1f us per launch)  -> %.2fx vs full" %
          (slab_rows, slab_rows * N * 4 / 1e6, nslab, us, us / nslab, full / us))
    # same with accumulate emulated by an extra C read+write per launch (torch add_)
    D = torch.zeros(M, N, device=dev)
    def run_acc():
        for sub_rp, sub_ci, sub_v in subs:
            spmm.csr_spmm(sub_rp, sub_ci, sub_v, B, out=C)
            D.add_(C)
    us2 = time_fn(run_acc, iters=2, warm=1)
    print("      with a separate accumulate pass per slab (upper bound on C traffic): %.0f us -> %.2fx" % (us2, full / us2))
    del subs
