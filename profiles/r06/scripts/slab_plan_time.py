"""The plan's column-slab tables (kernel="staged-slabs") against its other kernels on the dense community graph.
    python profiles/r06/scripts/slab_plan_time.py [graph] [slab counts ...]      (GESPMM_SLABS is read once per process: one count per run)"""
import os, statistics, sys
import torch
sys.path.insert(0, ".")
import gespmm_amd
from gespmm_amd import _lib, graphs, spmm

name = sys.argv[1] if len(sys.argv) > 1 else "reddit-sbm"
N = 128


def timed(fn, reps):
    for _ in range(2): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev)


g = graphs.synthetic_graph(name, seed=42, device="cuda")
rp, ci, M, K, nnz = g["rowptr"], g["colind"], g["M"], g["K"], g["nnz"]
val = torch.rand(nnz, device="cuda") - 0.5
B = torch.rand(K, N, device="cuda") - 0.5
C = torch.empty(M, N, device="cuda")
ref = spmm.csr_spmm(rp, ci, val, B, cfg={"flags": _lib.FLAG_STRICT_ORDER})
for kern in (("staged-slabs",) if os.environ.get("SLAB_ONLY") else ("auto", "seg-stream", "staged-slabs")):
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    plan = spmm.SpmmPlan(rp, ci, K, N, values=val, kernel=kern, reorder=True if os.environ.get("SLAB_REORDER") else "auto")
    torch.cuda.synchronize()
    pms = (time.perf_counter() - t0) * 1e3
    t = timed(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan), 7)
    same = bool(torch.equal(C.view(torch.int32), ref.view(torch.int32)))
    print("%s N=%d GESPMM_SLABS=%s kernel=%-12s %9.1f us  bits=%s  plan %.1f ms | %s" % (name, N, os.environ.get("GESPMM_SLABS", "-"), kern, t, same, pms,
          plan.describe().split("|")[-1].strip()[:160]), flush=True)
    del plan
