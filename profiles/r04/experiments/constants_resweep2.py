"""More of the older constants on the round-4 tree, stand-ins + hold-outs: (1) gather depth of clustered plans (4 vs 8 B rows in flight per lane
group: FLAG_SHALLOW_UNROLL forced / forbidden) at N = 32 / 64 / 128; (2) rows per wavefront of the PLAIN call's batch-stream kernel (round 1's
12 KB-per-task rule) at N = 32 / 128 / 512."""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
import gespmm_amd  # noqa
from gespmm_amd import graphs, spmm, _lib
import holdout_audit as ha
dev = torch.device("cuda")
def med(fn, n):
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev)
def cases():
    for n in ("com-amazon-sbm", "com-amazon-like"):
        yield n, (lambda n=n: graphs.synthetic_graph(n, seed=42, device=dev))
    yield "products-sbm x0.25", lambda: graphs.synthetic_graph("products-sbm", seed=42, device=dev, scale=0.25)
    yield "products-like x0.25", lambda: graphs.synthetic_graph("products-like", seed=42, device=dev, scale=0.25)
    for n, mk in ha.cases(["lfr-mu0.1", "lfr-mu0.3", "holme-kim-m5", "geometric", "nws-k10", "ba-m6"]):
        yield n, mk
for name, mk in cases():
    try:
        g = mk()
    except Exception as ex:
        print("%s skipped: %s" % (name, ex)); continue
    rp, ci, K, M, nnz = g["rowptr"], g["colind"], g["K"], g["M"], g["nnz"]
    val = torch.rand(nnz, device=dev) - 0.5
    n_it = 40 if nnz < 2e7 else 8
    for N in (32, 64, 128):
        B = torch.rand(K, N, device=dev) - 0.5
        C = torch.empty(M, N, device=dev)
        row = []
        for label, fl in (("auto", 0), ("U=4", _lib.FLAG_SHALLOW_UNROLL), ("U=8", 0x20000)):
            p = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True, kernel="stream", flags=fl)
            row.append("%s %.1f" % (label, med(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p), n_it)))
            del p
        print("%-20s mean %5.1f N=%-3d plan batch-stream depth: %s" % (name, nnz / M, N, "  ".join(row)), flush=True)
    for N in (32, 128, 512):
        B = torch.rand(K, N, device=dev) - 0.5
        C = torch.empty(M, N, device=dev)
        row = ["auto %.1f" % med(lambda: spmm.csr_spmm(rp, ci, val, B, out=C), n_it)]
        for rpw in (1, 2, 4, 8, 16, 32):
            try:
                row.append("rpw=%d %.1f" % (rpw, med(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, cfg={"rows_per_wave": rpw}), n_it)))
            except Exception as ex:
                row.append("rpw=%d n/a" % rpw)
        print("%-20s mean %5.1f N=%-3d plain call: %s | %s" % (name, nnz / M, N, "  ".join(row), _lib.lib and ""), flush=True)
    del g; torch.cuda.empty_cache()
