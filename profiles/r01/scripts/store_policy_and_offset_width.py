#!/usr/bin/env python3
"""Headline workload: store policy and offset-width knobs on the default kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gespmm_amd
from gespmm_amd import _lib as F, graphs, spmm

def time_fn(fn, iters=300, warm=30):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

dev = torch.device("cuda:0")
g = graphs.synthetic_graph("com-amazon-like", device=dev)
M, K, nnz = g["M"], g["K"], g["nnz"]
rp, ci = g["rowptr"], g["colind"]
val = torch.rand(nnz, device=dev)
for N in (128, 256):
    B = torch.rand(K, N, device=dev); C = torch.empty(M, N, device=dev)
    line = "N=%d:" % N
    for label, flags in (("default", 0), ("nt-store", F.FLAG_NT_STORE), ("no-xcd-remap", F.FLAG_NO_XCD_REMAP), ("u4", F.FLAG_SHALLOW_UNROLL),
                         ("nt+u4", F.FLAG_NT_STORE | F.FLAG_SHALLOW_UNROLL), ("idx64", F.FLAG_FORCE_IDX64)):
        line += " %s %.1f |" % (label, time_fn(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, cfg=dict(flags=flags))))
    print(line); sys.stdout.flush()
