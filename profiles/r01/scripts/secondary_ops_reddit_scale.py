#!/usr/bin/env python3
"""Secondary ops at reddit scale (115 M edges): csr2csc, SDDMM; checked against torch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gespmm_amd
from gespmm_amd import graphs, spmm, sddmm

def time_fn(fn, iters=3, warm=1):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "reddit-like"
g = graphs.synthetic_graph(name, device=dev)
M, K, nnz = g["M"], g["K"], g["nnz"]
rp, ci = g["rowptr"], g["colind"]
val = torch.rand(nnz, device=dev)
colptr = torch.empty(K + 1, dtype=torch.int32, device=dev); rowind = torch.empty(nnz, dtype=torch.int32, device=dev)
us = time_fn(lambda: spmm.csr2csc(rp, ci, colptr, rowind, val))
cv = spmm.csr2csc(rp, ci, colptr, rowind, val)
# check: (A^T) as CSR must reproduce A x = via both forms on a random vector block
x = torch.rand(K, 4, device=dev)
y1 = spmm.csr_spmm(colptr, rowind, cv, torch.rand(M, 4, device=dev) * 0 + 1.0)
rows = torch.repeat_interleave(torch.arange(M, device=dev), (rp[1:] - rp[:-1]).long())
y2 = torch.zeros(K, 4, device=dev).index_add_(0, ci.long(), val[:, None].expand(-1, 4).contiguous())
print("%s csr2csc: %.0f us ; column sums max rel err %.2e" % (name, us, ((y1 - y2).abs().max() / y2.abs().max()).item()))
for N in (41, 128):
    D1 = torch.rand(M, N, device=dev); D2 = torch.rand(K, N, device=dev)
    t_csr = time_fn(lambda: sddmm.csr_sddmm(rp, ci, D1, D2))
    t_coo = time_fn(lambda: sddmm.coo_sddmm(rows.int(), ci, D1, D2))
    o = sddmm.csr_sddmm(rp, ci, D1, D2)
    idx = torch.randint(0, nnz, (100000,), device=dev)
    ref = (D1[rows[idx]] * D2[ci[idx].long()]).sum(1)
    print("  sddmm N=%d: csr %.0f us, coo %.0f us, sampled max rel err %.2e" % (N, t_csr, t_coo, ((o[idx] - ref).abs().max() / ref.abs().max()).item()))
