#!/usr/bin/env python3
"""Launch-bound loops: L back-to-back products through plans, enqueued eagerly (Python -> pybind -> hipLaunchKernel per product) against ONE
hipGraph replay of the same L launches (torch.cuda.CUDAGraph). pubmed-sized graph (the reference's GCN benchmark: kernels of 13-18 us,
the epoch loop is host-bound — profiles/r06/gcn_epochs.log) and the headline graph at N = 32.
    python profiles/r06/scripts/graph_replay_timing.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from gespmm_amd import graphs, spmm  # noqa: E402

dev = torch.device("cuda")
for name, N, L in (("pubmed-like", 128, 20), ("pubmed-like", 32, 20), ("com-amazon-sbm", 32, 20), ("com-amazon-sbm", 128, 20)):
    g = graphs.synthetic_graph(name, seed=42, device=dev)
    rp, ci, K, M, nnz = g["rowptr"], g["colind"], g["K"], g["M"], g["nnz"]
    val = torch.rand(nnz, device=dev) - 0.5
    plan = spmm.SpmmPlan(rp, ci, K, N, values=val, expected_launches=1000000)
    B = torch.rand(K, N, device=dev) - 0.5
    C = torch.empty((M, N), device=dev)

    def loop():
        for _ in range(L):
            spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan)

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        loop()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loop()
    ref = C.clone()

    def wall(fn, reps=50):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps / L * 1e6

    t_eager, t_graph = wall(loop), wall(graph.replay)
    graph.replay()
    torch.cuda.synchronize()
    same = torch.equal(C.view(torch.int32), ref.view(torch.int32))
    print("%-15s N=%-3d %d launches per loop: eager %.1f us per product (wall), one graph replay %.1f us per product  x%.2f  bits %s | %s" % (
        name, N, L, t_eager, t_graph, t_eager / t_graph, "same" if same else "DIFFER", plan.describe().split("|")[-1].strip()[:40]), flush=True)
