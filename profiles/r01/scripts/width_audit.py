#!/usr/bin/env python3
"""Width audit: AUTO over every kind of N (odd, even, multiples of 4, around tile edges) on one graph —
time per column should move smoothly; spikes point at a bad geometry choice."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gespmm_amd
from gespmm_amd import graphs, spmm

def time_fn(fn, iters=100, warm=10):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "com-amazon-like"
g = graphs.synthetic_graph(name, device=dev)
rp, ci, M, K = g["rowptr"], g["colind"], g["M"], g["K"]
val = torch.rand(ci.numel(), device=dev)
prev = None
NS = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else None
IT = 3 if ci.numel() > 2e7 else 100
for N in NS or (1, 2, 3, 4, 6, 8, 12, 16, 17, 24, 31, 32, 33, 48, 63, 64, 65, 66, 68, 96, 100, 127, 128, 129, 130, 132, 160, 192, 200, 255, 256, 257, 260, 320, 384, 500, 512, 513, 516, 768, 1024):
    B = torch.rand(K, N, device=dev); C = torch.empty(M, N, device=dev)
    us = time_fn(lambda: spmm.csr_spmm(rp, ci, val, B, out=C), IT, 1 if IT < 10 else 10)
    per = us / N
    note = ""
    if prev is not None and per > 1.35 * prev[1] and N > 8:
        note = "  <-- %.0f%% more per column than N=%d" % (100 * (per / prev[1] - 1), prev[0])
    print("N=%4d  %8.1f us  %.3f us/col%s" % (N, us, per, note)); sys.stdout.flush()
    prev = (N, per)
    del B, C
