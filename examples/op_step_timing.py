#!/usr/bin/env python3
"""One aggregation layer of BASELINE.json's config 4 at op level: SPMMFunction forward (SpMM) and
backward (SpMM on the CSC arrays for the features, SDDMM for the edge-weight gradient), hidden = 128.

    python examples/op_step_timing.py [pubmed|reddit-like|com-amazon-like] [N]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import gespmm_amd  # noqa: E402,F401
from gespmm_amd import SPMMFunction, graphs, spmm  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "pubmed"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    dev = torch.device("cuda:0")
    if name == "pubmed":
        g = graphs.load_mtx_as_csr(os.path.join(ROOT, "tests", "golden", "pubmed.mtx"))
        rp = torch.as_tensor(g["rowptr"]).to(dev)
        ci = torch.as_tensor(g["colind"]).to(dev)
        M = g["M"]
    else:
        g = graphs.synthetic_graph(name, device=dev)
        rp, ci, M = g["rowptr"], g["colind"], g["M"]
    nnz = ci.numel()
    w_csr = torch.rand(nnz, device=dev, requires_grad=True)
    colptr = torch.empty(M + 1, dtype=torch.int32, device=dev)
    rowind = torch.empty(nnz, dtype=torch.int32, device=dev)
    w_csc = spmm.csr2csc(rp, ci, colptr, rowind, w_csr.detach())
    x = torch.rand(M, N, device=dev, requires_grad=True)
    go = torch.rand(M, N, device=dev)

    def step(edge_grad):
        y = SPMMFunction.apply(rp, ci, colptr, rowind, x, w_csr, w_csc, edge_grad)
        y.backward(go)
        x.grad = None
        w_csr.grad = None

    for edge_grad in (False, True):
        for _ in range(3):
            step(edge_grad)
        iters = 5 if nnz > 2e7 else 100
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            step(edge_grad)
        e1.record()
        torch.cuda.synchronize()
        print("%s N=%d nnz=%d: forward + backward%s  %.3f ms" % (
            name, N, nnz, " + edge-weight gradient (SDDMM)" if edge_grad else "", e0.elapsed_time(e1) / iters))


if __name__ == "__main__":
    main()
