"""The plan's staged-rows kernel against its streaming kernels.   python scripts/staged_time.py [graph ...] [--n=128,256]"""
import statistics, sys
import torch
sys.path.insert(0, ".")
import gespmm_amd
from gespmm_amd import graphs, spmm

names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["products-sbm", "com-amazon-sbm", "products-like", "reddit-like"]
widths = [128, 256]
for a in sys.argv[1:]:
    if a.startswith("--n="): widths = [int(x) for x in a[4:].split(",")]


def timed(fn, reps):
    for _ in range(2): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev)


for name in names:
    g = graphs.synthetic_graph(name, seed=42, device="cuda")
    rp, ci, M, K, nnz = g["rowptr"], g["colind"], g["M"], g["K"], g["nnz"]
    val = torch.rand(nnz, device="cuda") - 0.5
    reps = 5 if nnz > 5e7 else 50
    for N in widths:
        B = torch.rand(K, N, device="cuda") - 0.5
        C = torch.empty(M, N, device="cuda")
        res = {}
        for kern in ("auto", "stream", "seg-stream", "staged"):
            try:
                plan = spmm.SpmmPlan(rp, ci, K, N, values=val, kernel=kern)
            except Exception as ex:
                print(name, N, kern, "plan failed:", str(ex)[:80]); continue
            d = plan.describe()
            t = timed(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan), reps)
            if "ref" not in res: res["ref"] = C.clone()
            same = bool(torch.equal(C.view(torch.int32), res["ref"].view(torch.int32)))
            print("%-16s N=%3d kernel=%-10s %9.1f us %7.1f GF/s bits=%s | %s" % (name, N, kern, t, 2.0 * nnz * N / t / 1e3, same,
                  d.split("|")[-1].strip()[:110] + (" analysis=" + d.split("analysis=")[1].split()[0] if "analysis=" in d else "")), flush=True)
            del plan
