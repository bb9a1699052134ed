"""A reddit-SIZED graph with planted communities (232 965 rows, 114.6 M entries, mean degree 492; communities of ~800 rows with ~330 of a row's entries
inside, ids shuffled): the cache-blocked path the library takes for dense graphs against a clustered plan on the streaming / staged-rows kernels.
    python scripts/dense_community_time.py [N ...]"""
import statistics, sys, time
import torch
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


M, nnz = 232965, 114615892
t0 = time.time()
rp, ci, _ = graphs.community_csr(M, nnz, 290, 16, 330.0, 0.6, 1.5, 1.55, 42, "cuda")
torch.cuda.synchronize()
deg = (rp[1:] - rp[:-1])
print("generated in %.1f s: M %d nnz %d mean degree %.0f longest row %d" % (time.time() - t0, M, int(ci.numel()), float(deg.float().mean()), int(deg.max())), flush=True)
val = torch.rand(int(ci.numel()), device="cuda") - 0.5
for N in [int(x) for x in sys.argv[1:]] or [128, 256]:
    B = torch.rand(M, N, device="cuda") - 0.5
    C = torch.empty(M, N, device="cuda")
    t_plain = timed(lambda: spmm.csr_spmm(rp, ci, val, B, out=C))
    ref = C.clone()
    print("N=%3d plain call %9.1f us" % (N, t_plain), flush=True)
    for kw in (dict(), dict(reorder=True, flags=0x800), dict(reorder=True, flags=0x800, kernel="stream"), dict(reorder=True, flags=0x800, kernel="seg-stream"), dict(reorder=True, flags=0x800, kernel="staged")):
        t1 = time.time()
        plan = spmm.SpmmPlan(rp, ci, M, N, values=val, **kw)
        torch.cuda.synchronize(); dt = time.time() - t1
        t = timed(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan))
        err = float((C - ref).abs().max())
        same = bool(torch.equal(C.view(torch.int32), ref.view(torch.int32)))
        print("N=%3d plan %-40s %9.1f us  x%.2f vs plain  bits=%s (max |diff| %.2g)  plan %.2f s | %s" % (N, str(kw), t, t_plain / t, same, err, dt, plan.describe()[:60] + " ... " + plan.describe().split("|")[-1].strip()[:100]), flush=True)
        del plan
