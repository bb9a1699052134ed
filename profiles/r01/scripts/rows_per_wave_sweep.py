#!/usr/bin/env python3
"""Headline workload (com-amazon-like, N=128, valued): rows per lane group / per wavefront for both
stream kernels, measured with many iterations (the differences are a few per cent)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gespmm_amd
from gespmm_amd import _lib as F, graphs, spmm

ITERS = int(sys.argv[3]) if len(sys.argv) > 3 else 200
def time_fn(fn, iters=ITERS, warm=max(2, ITERS // 10)):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "com-amazon-like"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
LOC = float(os.environ.get("LOCALITY", "0"))
g = graphs.synthetic_graph(name, device=dev, locality=LOC)
M, K, nnz = g["M"], g["K"], g["nnz"]
rp, ci = g["rowptr"], g["colind"]
val = torch.rand(nnz, device=dev)
UNW = os.environ.get("UNWEIGHTED") == "1"
def call(cfg=None):
    if UNW:
        return spmm.csr_spmm_no_edge_value(rp, ci, B, out=C, cfg=cfg)
    return spmm.csr_spmm(rp, ci, val, B, out=C, cfg=cfg)
B = torch.rand(K, N, device=dev); C = torch.empty(M, N, device=dev)
print("%s N=%d auto: %.1f us" % (name, N, time_fn(lambda: call())))
for kname, flag in (("seg", F.FLAG_SEG_STREAM), ("batch", F.FLAG_BATCH_STREAM)):
    line = "  %-5s:" % kname
    for r in (1, 2, 4, 8, 16):
        for extra, tag in ((0, ""),):
            cfg = dict(rows_per_wave=r, flags=flag | extra)
            line += " r%d%s %.1f |" % (r, tag, time_fn(lambda: call(cfg)))
    print(line); sys.stdout.flush()
