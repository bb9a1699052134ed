#!/usr/bin/env python3
"""Host-side cost of one op call (Python/ctypes binding) on a tiny graph."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gespmm_amd
from gespmm_amd import graphs, spmm, SPMMFunction
dev = torch.device("cuda:0")
g = graphs.load_mtx_as_csr(os.path.join(ROOT, "tests", "golden", "cora.mtx"))
rp = torch.from_numpy(g["rowptr"]).to(dev); ci = torch.from_numpy(g["colind"]).to(dev)
B = torch.rand(g["K"], 16, device=dev); out = torch.empty(g["M"], 16, device=dev)
def bench(fn, n=3000):
    for _ in range(200): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    host = (time.perf_counter() - t) / n * 1e6
    torch.cuda.synchronize(); tot = (time.perf_counter() - t) / n * 1e6
    return host, tot
print("csr_spmm_no_edge_value(out=)   host %.1f us/call, total %.1f us/call" % bench(lambda: spmm.csr_spmm_no_edge_value(rp, ci, B, out=out)))
print("csr_spmm_no_edge_value()       host %.1f us/call, total %.1f us/call" % bench(lambda: spmm.csr_spmm_no_edge_value(rp, ci, B)))
print("torch.empty only               host %.1f us/call, total %.1f us/call" % bench(lambda: torch.empty((g["M"], 16), device=dev)))
print("B + 1 (one eager torch op)     host %.1f us/call, total %.1f us/call" % bench(lambda: B + 1))
colptr, rowind = graphs.transpose_csr(rp, ci)
x = B.clone().requires_grad_(True)
def fb():
    y = SPMMFunction.apply(rp, ci, colptr, rowind, x); y.sum().backward()
print("SPMMFunction fwd+sum+bwd        host %.1f us/call, total %.1f us/call" % bench(fb, 1000))
