"""N <= 64: V = 1 (one lane per column: 2 rows per gather instruction at N = 32) against V = 4 (8 lanes per 32 columns: 8 rows per
instruction) by mean degree — plain call and clustered plan, stand-ins and hold-out graphs. The rule for select.cpp: auto_variant."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch

import gespmm_amd  # noqa
from gespmm_amd import graphs, spmm
import holdout_audit as ha

dev = torch.device("cuda")


def med(fn, n=40):
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


def all_cases():
    for n in ("com-amazon-sbm", "com-amazon-like", "cit-hepth-like", "pubmed-like"):
        yield n, (lambda n=n: graphs.synthetic_graph(n, seed=42, device=dev))
    yield "products-sbm x0.25", (lambda: graphs.synthetic_graph("products-sbm", seed=42, device=dev, scale=0.25))
    yield "products-like x0.25", (lambda: graphs.synthetic_graph("products-like", seed=42, device=dev, scale=0.25))
    for n, mk in ha.cases([]):
        yield n, mk


for name, make in all_cases():
    try:
        g = make()
    except Exception as ex:  # noqa: BLE001
        print("== %s skipped: %s" % (name, ex))
        continue
    M, K, nnz = g["M"], g["K"], g["nnz"]
    rp, ci = g["rowptr"], g["colind"]
    val = torch.rand(nnz, device=dev) - 0.5
    for N in (16, 32, 64):
        B = torch.rand(K, N, device=dev) - 0.5
        C = torch.empty(M, N, device=dev)
        n_it = 40 if nnz < 2e7 else 10
        row = []
        ref = None
        for variant in (1, 2, 3):
            t_plain = med(lambda: spmm.csr_spmm(rp, ci, val, B, variant=variant, out=C), n_it)
            if ref is None:
                ref = C.clone()
            ok = torch.equal(C.view(torch.int32), ref.view(torch.int32))
            p = spmm.SpmmPlan(rp, ci, K, N, variant=variant, values=val)
            t_plan = med(lambda: spmm.csr_spmm(rp, ci, val, B, variant=variant, out=C, plan=p), n_it)
            ok = ok and torch.equal(C.view(torch.int32), ref.view(torch.int32))
            kern = "seg" if "segmented" in p.describe() else "bat"
            row.append("V=%d plain %8.1f plan %8.1f (%s%s)%s" % ({1: 1, 2: 2, 3: 4}[variant], t_plain, t_plan, "clu/" if p.clustered else "sto/", kern,
                                                                 "" if ok else " BITS?"))
            del p
        print("%-38s mean deg %6.1f N=%-2d %s" % (name, nnz / M, N, " | ".join(row)), flush=True)
    del g
    torch.cuda.empty_cache()
