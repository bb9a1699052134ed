"""Staged-rows kernel at N = 64 (one dword per lane, 256 staged rows per block): against the streaming kernels of the same plan, block
height swept (GESPMM_STAGED_ROWS64 is read once per process: one subprocess per height)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
child = r'''
import sys, statistics, torch
sys.path.insert(0, %r)
import gespmm_amd
from gespmm_amd import graphs, spmm
dev = "cuda"
def med(fn, n):
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev)
for name, scale in (("products-sbm", 1.0), ("com-amazon-sbm", 1.0), ("reddit-sbm", 1.0)):
    g = graphs.synthetic_graph(name, seed=42, device=dev, scale=scale)
    rp, ci, K, M, nnz = g["rowptr"], g["colind"], g["K"], g["M"], g["nnz"]
    val = torch.rand(nnz, device=dev) - 0.5
    B = torch.rand(K, 64, device=dev) - 0.5
    C = torch.empty(M, 64, device=dev)
    n = 50 if nnz < 1e7 else 8
    ref = spmm.csr_spmm(rp, ci, val, B)
    row = []
    for kern in ("stream", "seg-stream", "staged"):
        try:
            p = spmm.SpmmPlan(rp, ci, K, 64, values=val, reorder=True, kernel=kern)
        except Exception as ex:
            row.append("%%s: %%s" %% (kern, str(ex)[:60])); continue
        t = med(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p), n)
        ok = torch.equal(C.view(torch.int32), ref.view(torch.int32))
        d = p.describe()
        row.append("%%s %%.1f us%%s%%s" %% (kern, t, "" if ok else " BITS!", (" (share " + d.split("staged_entries=")[1].split(" ")[0] + ")") if "staged_entries=" in d else ""))
        del p
    print("%%-16s N=64 rows/block=%%s  %%s" %% (name, sys.argv[1], "  ".join(row)), flush=True)
    del g
    torch.cuda.empty_cache()
''' % ROOT
for r in ("64", "128", "192", "256"):
    env = dict(os.environ, GESPMM_STAGED_ROWS64=r)
    out = subprocess.run([sys.executable, "-c", child, r], env=env, capture_output=True, text=True)
    print(out.stdout.strip() or out.stderr[-600:], flush=True)
