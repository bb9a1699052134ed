import sys, statistics, torch
sys.path.insert(0, ".")
import gespmm_amd
from gespmm_amd import graphs, spmm
def timed(fn, reps=7):
    for _ in range(2): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev)
M, nnz = graphs.SPECS["reddit-like"][:2]
rp, ci, _ = graphs.community_csr(M, nnz, 290, 16, 330.0, 0.6, 1.5, 1.55, 42, "cuda")
val = torch.rand(int(ci.numel()), device="cuda") - 0.5
for N in (16, 32, 64, 128, 256, 512):
    B = torch.rand(M, N, device="cuda") - 0.5
    C = torch.empty(M, N, device="cuda")
    out = []
    for kern in ("auto", "stream", "seg-stream"):
        p = spmm.SpmmPlan(rp, ci, M, N, values=val, kernel=kern)
        t = timed(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p))
        out.append("%s %8.1f us (%s)" % (kern, t, p.describe().split("kernel=")[1].split()[0]))
        del p
    print("reddit-sbm N=%3d: " % N + " | ".join(out), flush=True)
