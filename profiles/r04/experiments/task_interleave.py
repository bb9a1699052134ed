"""Do the XCDs finish together? The planned kernels give XCD x the x-th contiguous eighth of the task table. Here the table is re-dealt
so that XCDs take interleaved chunks of CH workgroups (GESPMM_TASK_INTERLEAVE): if the spread between orders of the same quality
(cluster_levels.log: +-5 % on the structureless graph) is one XCD's slice running long, interleaving evens it out."""
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


for name, scale, widths in [("com-amazon-sbm", 1.0, (128, 32, 512)), ("com-amazon-like", 1.0, (128, 32, 512)), ("products-sbm", 0.25, (128, 32))]:
    g = graphs.synthetic_graph(name, seed=42, device=dev, scale=scale)
    rp, ci, K, M, nnz = g["rowptr"], g["colind"], g["K"], g["M"], g["nnz"]
    val = torch.rand(nnz, device=dev) - 0.5
    for N in widths:
        B = torch.rand(K, N, device=dev) - 0.5
        C = torch.empty(M, N, device=dev)
        ref = None
        for lv in (3, 5, 6):
            row = []
            for ch in (0, 8, 32, 128, 512):
                os.environ["GESPMM_CLUSTER_LEVELS"] = str(lv)
                os.environ["GESPMM_TASK_INTERLEAVE"] = str(ch)
                p = spmm.SpmmPlan(rp, ci, K, N, values=val, kernel="stream")
                t = med(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p), 100 if nnz < 1e7 else 10)
                if ref is None:
                    ref = C.clone()
                ok = torch.equal(C.view(torch.int32), ref.view(torch.int32))
                row.append("CH=%d %.1f%s" % (ch, t, "" if ok else " BITS!"))
                del p
            print("%-16s N=%-3d levels=%d  %s" % (name, N, lv, "  ".join(row)), flush=True)
        del os.environ["GESPMM_CLUSTER_LEVELS"], os.environ["GESPMM_TASK_INTERLEAVE"]
