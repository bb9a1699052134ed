"""`nt` gathers for far columns in the staged-rows kernel, on / off (GESPMM_STAGED_FAR_BLOCKS = 64 / 0), on every graph where the kernel is
taken: which graphs want them? (products-shaped communities gained 12 % in round 3; the headline graph LOSES 5-9 % at N = 256 / 512.)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
child = r'''
import sys, statistics, torch
sys.path.insert(0, %r); sys.path.insert(0, %r + "/scripts")
import gespmm_amd
from gespmm_amd import graphs, spmm
import holdout_audit as ha
dev = "cuda"
def med(fn, n):
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev)
def cases():
    yield "com-amazon-sbm", lambda: graphs.synthetic_graph("com-amazon-sbm", seed=42, device=dev)
    yield "products-sbm x0.25", lambda: graphs.synthetic_graph("products-sbm", seed=42, device=dev, scale=0.25)
    yield "products-sbm", lambda: graphs.synthetic_graph("products-sbm", seed=42, device=dev)
    for n, mk in ha.cases(["geometric", "nws-k10", "lfr-mu0.1"]):
        yield n, mk
for name, mk in cases():
    try:
        g = mk()
    except Exception as ex:
        print("%%s skipped %%s" %% (name, ex)); continue
    rp, ci, K, M, nnz = g["rowptr"], g["colind"], g["K"], g["M"], g["nnz"]
    val = torch.rand(nnz, device=dev) - 0.5
    row = []
    for N in (128, 256, 512):
        B = torch.rand(K, N, device=dev) - 0.5
        C = torch.empty(M, N, device=dev)
        p = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True, kernel="staged")
        t = med(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p), 50 if nnz < 2e7 else 8)
        row.append("N=%%d %%.1f" %% (N, t))
        del p, B, C
    print("%%-20s far=%%-3s B(N=128)=%%5.0f MB mean degree %%5.1f  %%s" %% (name, sys.argv[1], K * 512 / 1e6, nnz / M, "  ".join(row)), flush=True)
    del g; torch.cuda.empty_cache()
''' % (ROOT, ROOT)
for far in ("0", "64", "0", "64"):
    out = subprocess.run([sys.executable, "-c", child, far], env=dict(os.environ, GESPMM_STAGED_FAR_BLOCKS=far), capture_output=True, text=True)
    print(out.stdout.strip() or out.stderr[-500:], flush=True)
