#!/usr/bin/env python3
"""The round-1 review's second yardstick: the BENCH graph's rows with B resident in L2 (columns folded into K = 2048 rows of B):
AUTO / batch / segmented kernels, plain and system-scope stores."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch

import gespmm_amd  # noqa: F401,E402
from gespmm_amd import graphs, spmm  # noqa: E402

dev = torch.device("cuda")
g = graphs.synthetic_graph("com-amazon-like", seed=42, device=dev)
M, nnz = g["M"], g["nnz"]
rp = g["rowptr"]
val = torch.rand(nnz, device=dev) - 0.5
N = 128


def timeit(fn, iters=200):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for K in (2048, 8192, 65536):
    ci = (g["colind"] % K).to(torch.int32).contiguous()
    B = torch.rand((K, N), device=dev)
    C = torch.empty((M, N), device=dev)
    row = []
    for lab, fl in (("auto", 0), ("batch", 0x20), ("seg", 0x80), ("auto sc1", 0x8000), ("batch sc1", 0x8020), ("seg sc1", 0x8080), ("seg U4", 0x90), ("seg U4 sc1", 0x8090)):
        us = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, cfg={"flags": fl} if fl else None))
        row.append("%s %.1f" % (lab, us))
    print("K=%-6d (B %.1f MB): %s" % (K, K * N * 4 / 1e6, " | ".join(row)), flush=True)
