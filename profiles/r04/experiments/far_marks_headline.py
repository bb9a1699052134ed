"""What do `nt` gathers for far columns buy on the HEADLINE graph? The staged-rows kernel has the marks (GESPMM_STAGED_FAR_BLOCKS; the
streaming kernels have none): forced staged plans at N = 128 / 256 / 512 with the marks off (0) and at several distances. If the marks
buy nothing here, carrying them into the streaming kernels (a per-entry flag + two load instructions under EXEC masks) is not worth it."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
child = r'''
import sys, statistics, torch
sys.path.insert(0, %r)
import gespmm_amd
from gespmm_amd import graphs, spmm
dev = "cuda"
def med(fn, n):
    for _ in range(5): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev)
for name in ("com-amazon-sbm", "com-amazon-like"):
    g = graphs.synthetic_graph(name, seed=42, device=dev)
    rp, ci, K, M, nnz = g["rowptr"], g["colind"], g["K"], g["M"], g["nnz"]
    val = torch.rand(nnz, device=dev) - 0.5
    row = []
    for N in (128, 256, 512):
        B = torch.rand(K, N, device=dev) - 0.5
        C = torch.empty(M, N, device=dev)
        p = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True, kernel="staged")
        t = med(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p), 100)
        ps = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True, kernel="stream")
        ts = med(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=ps), 100)
        row.append("N=%%d staged %%.1f (batch-stream %%.1f)" %% (N, t, ts))
    print("%%-16s far=%%s  %%s" %% (name, sys.argv[1], "  ".join(row)), flush=True)
''' % ROOT
for far in ("0", "8", "64", "512"):
    out = subprocess.run([sys.executable, "-c", child, far], env=dict(os.environ, GESPMM_STAGED_FAR_BLOCKS=far), capture_output=True, text=True)
    print(out.stdout.strip() or out.stderr[-500:], flush=True)
