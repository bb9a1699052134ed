#!/usr/bin/env python3
"""SDDMM lane-group width sweep (GESPMM_EXP_SDW is read by an experiment build only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gespmm_amd
from gespmm_amd import graphs, sddmm

def time_fn(fn, iters=3, warm=1):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

dev = torch.device("cuda:0")
for name in sys.argv[1].split(","):
    g = graphs.synthetic_graph(name, device=dev)
    M, K, nnz = g["M"], g["K"], g["nnz"]
    rp, ci = g["rowptr"], g["colind"]
    rows = torch.repeat_interleave(torch.arange(M, device=dev), (rp[1:] - rp[:-1]).long()).int()
    it = 3 if nnz > 2e7 else 50
    for N in (16, 32, 41, 64, 128, 256):
        D1 = torch.rand(M, N, device=dev); D2 = torch.rand(K, N, device=dev)
        line = "%-16s N=%3d:" % (name, N)
        for W in ("", "4", "8", "16", "32", "64"):
            if W: os.environ["GESPMM_EXP_SDW"] = W
            else: os.environ.pop("GESPMM_EXP_SDW", None)
            line += " W%s csr %.0f coo %.0f |" % (W or "auto", time_fn(lambda: sddmm.csr_sddmm(rp, ci, D1, D2), it), time_fn(lambda: sddmm.coo_sddmm(rows, ci, D1, D2), it))
        print(line); sys.stdout.flush()
