#!/usr/bin/env python3
"""Wide feature widths on the big graphs: variant 3 (one strip per lane, column tiles over workgroups) vs variant 4 (two strips)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch

import gespmm_amd  # noqa: F401,E402
from gespmm_amd import graphs, spmm  # noqa: E402

dev = torch.device("cuda")


def timeit(fn, iters):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name in ("products-like", "com-amazon-like"):
    g = graphs.synthetic_graph(name, seed=42, device=dev)
    M, K, nnz = g["M"], g["K"], g["nnz"]
    rp, ci = g["rowptr"], g["colind"]
    val = torch.rand(nnz, device=dev) - 0.5
    for N in (256, 512, 1024):
        if name == "products-like" and N > 512:
            continue
        B = torch.rand((K, N), device=dev)
        C = torch.empty((M, N), device=dev)
        row = []
        for lab, variant, cfg in (("auto", -1, None), ("v3", 3, None), ("v4", 4, None), ("v3 U4", 3, {"flags": 0x10}), ("v4 rpw2", 4, {"rows_per_wave": 2})):
            try:
                us = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, variant=variant, cfg=cfg, out=C), 3 if nnz > 5e7 else 50)
                row.append("%s %.0f" % (lab, us))
            except Exception as ex:  # noqa: BLE001
                row.append("%s failed" % lab)
        print("%-16s N=%-5d %s" % (name, N, " | ".join(row)), flush=True)
