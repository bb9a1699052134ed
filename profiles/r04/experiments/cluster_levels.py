"""Clustering depth (GESPMM_CLUSTER_LEVELS) A/B in ONE process on one box: plans with 2..8 levels on the stand-ins, two interleaved
rounds of per-launch event medians — round 3 cut the default from 6 to 3 levels, the structureless stand-in lost 4.5 %."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch

import gespmm_amd  # noqa
from gespmm_amd import graphs, spmm

dev = torch.device("cuda")


def med(fn, n):
    for _ in range(5):
        fn()
    s = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    e = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    torch.cuda.synchronize()
    for i in range(n):
        s[i].record()
        fn()
        e[i].record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) * 1e3 for a, b in zip(s, e))


cases = [("com-amazon-sbm", 1.0, (128, 32, 512)), ("com-amazon-like", 1.0, (128, 32, 512)), ("products-sbm", 0.25, (128, 32)),
         ("products-sbm", 1.0, (128,))]
for name, scale, widths in cases:
    g = graphs.synthetic_graph(name, seed=42, device=dev, scale=scale)
    rp, ci, K, M, nnz = g["rowptr"], g["colind"], g["K"], g["M"], g["nnz"]
    val = torch.rand(nnz, device=dev) - 0.5
    for N in widths:
        B = torch.rand(K, N, device=dev) - 0.5
        C = torch.empty(M, N, device=dev)
        plans = {}
        for lv in (2, 3, 4, 5, 6, 8):
            os.environ["GESPMM_CLUSTER_LEVELS"] = str(lv)
            plans[lv] = spmm.SpmmPlan(rp, ci, K, N, values=val, kernel="stream" if N != 32 else "auto")
        del os.environ["GESPMM_CLUSTER_LEVELS"]
        n = 100 if nnz < 1e7 else 10
        rounds = []
        for _ in range(2):
            rounds.append({lv: med(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p), n) for lv, p in plans.items()})
        print("%-16s x%.2f N=%-3d " % (name, scale, N) + " | ".join(
            "L%d %.1f %.1f (%s, model %s)" % (lv, rounds[0][lv], rounds[1][lv], plans[lv].describe().split("clusters=")[1].split(" ")[0].split(">")[-1],
                                              plans[lv].describe().split("->")[1].split(" ")[0]) for lv in plans), flush=True)
        del plans
    del g
    torch.cuda.empty_cache()
