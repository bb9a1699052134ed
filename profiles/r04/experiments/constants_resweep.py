"""Round 3's constants on the round-4 tree (the far marks turned out stale: far_marks_by_graph.log): (1) rows per block of the staged-rows kernel
(GESPMM_STAGED_BLOCK_ROWS, experiment knob, one process per setting), (2) non-zeros per wavefront task of the batch-stream kernel (plan option)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
common = r'''
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
''' % (ROOT, ROOT)
staged_child = common + r'''
def cases():
    yield "com-amazon-sbm", lambda: graphs.synthetic_graph("com-amazon-sbm", seed=42, device=dev)
    yield "products-sbm x0.25", lambda: graphs.synthetic_graph("products-sbm", seed=42, device=dev, scale=0.25)
    yield "products-sbm", lambda: graphs.synthetic_graph("products-sbm", seed=42, device=dev)
    for n, mk in ha.cases(["geometric", "nws-k10"]):
        yield n, mk
for name, mk in cases():
    g = mk()
    rp, ci, K, M, nnz = g["rowptr"], g["colind"], g["K"], g["M"], g["nnz"]
    val = torch.rand(nnz, device=dev) - 0.5
    row = []
    for N in (128, 256, 512):
        B = torch.rand(K, N, device=dev) - 0.5
        C = torch.empty(M, N, device=dev)
        p = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True, kernel="staged")
        t = med(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p), 40 if nnz < 2e7 else 8)
        row.append("N=%d %.1f (share %s)" % (N, t, p.describe().split("staged_entries=")[1].split(" ")[0]))
        del p, B, C
    print("%-20s rows/block=%-3s  %s" % (name, sys.argv[1], "  ".join(row)), flush=True)
    del g; torch.cuda.empty_cache()
'''
task_child = common + r'''
def cases():
    for n in ("com-amazon-sbm", "com-amazon-like"):
        yield n, (lambda n=n: graphs.synthetic_graph(n, seed=42, device=dev))
    yield "products-sbm x0.25", lambda: graphs.synthetic_graph("products-sbm", seed=42, device=dev, scale=0.25)
    for n, mk in ha.cases(["lfr-mu0.1", "lfr-mu0.3", "holme-kim-m5", "geometric"]):
        yield n, mk
for name, mk in cases():
    g = mk()
    rp, ci, K, M, nnz = g["rowptr"], g["colind"], g["K"], g["M"], g["nnz"]
    val = torch.rand(nnz, device=dev) - 0.5
    for N, tes in ((32, (0, 48, 64, 96, 128, 192, 256)), (128, (0, 24, 32, 40, 48, 64, 96, 128)), (512, (0, 16, 24, 32, 48, 64, 96))):
        B = torch.rand(K, N, device=dev) - 0.5
        C = torch.empty(M, N, device=dev)
        row = []
        for te in tes:
            p = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True, kernel="stream", task_entries=te)
            t = med(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p), 40 if nnz < 2e7 else 8)
            row.append("%s %.1f" % ("default(%s)" % p.describe().split("task_entries=")[1].split(" ")[0] if te == 0 else "te=%d" % te, t))
            del p
        print("%-20s N=%-3d batch-stream  %s" % (name, N, "  ".join(row)), flush=True)
        del B, C
    del g; torch.cuda.empty_cache()
'''
for r in ("48", "64", "80", "96", "128", "160"):
    out = subprocess.run([sys.executable, "-c", staged_child, r], env=dict(os.environ, GESPMM_STAGED_BLOCK_ROWS=r), capture_output=True, text=True)
    print(out.stdout.strip() or out.stderr[-500:], flush=True)
out = subprocess.run([sys.executable, "-c", task_child], capture_output=True, text=True)
print(out.stdout.strip() or out.stderr[-800:], flush=True)
