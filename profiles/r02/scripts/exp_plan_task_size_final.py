import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import gespmm_amd
from gespmm_amd import graphs, spmm
dev = torch.device("cuda")
def timeit(fn, iters=200):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name in ("com-amazon-sbm", "com-amazon-like"):
    g = graphs.synthetic_graph(name, seed=42, device=dev)
    M, K, nnz = g["M"], g["K"], g["nnz"]; rp, ci = g["rowptr"], g["colind"]
    val = torch.rand(nnz, device=dev) - 0.5
    for N in (64, 128, 256):
        B = torch.rand((K, N), device=dev); C = torch.empty((M, N), device=dev)
        for flags, lab in ((0x20000, "U8"), (0x10, "U4")):
            row = []
            for te in (24, 32, 40, 48, 56, 64, 80, 96, 128):
                plan = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True, task_entries=te, kernel="stream", flags=flags)
                row.append("%d:%.1f" % (te, timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan))))
            print("%s N=%d stream %s  %s" % (name, N, lab, " ".join(row)), flush=True)
