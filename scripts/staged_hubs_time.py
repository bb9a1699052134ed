"""The staged-rows kernel on a community graph WITH hub rows: the products-shaped community stand-in at 1/4 size, `nhub` of its rows given 3 000 ... 20 000
extra random entries (the real ogbn-products has rows up to 17 k entries; the stand-in stops at 1 446). Staged-rows (hubs handed to the long-row pass)
against the streaming kernels of the same plan.   python scripts/staged_hubs_time.py [nhub ...]"""
import statistics, sys
import torch
sys.path.insert(0, ".")
import gespmm_amd
from gespmm_amd import graphs, spmm


def timed(fn, reps=20):
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev)


g = graphs.synthetic_graph("products-sbm", seed=42, device="cuda", scale=0.25)
M, K = g["M"], g["K"]
gen = torch.Generator(device="cuda"); gen.manual_seed(5)
for nhub in [int(x) for x in sys.argv[1:]] or [0, 20, 200]:
    rp, ci = g["rowptr"].to(torch.int64), g["colind"]
    if nhub:
        rows = torch.randperm(M, generator=gen, device="cuda")[:nhub]
        extra = torch.randint(3000, 20000, (nhub,), generator=gen, device="cuda")
        deg = (rp[1:] - rp[:-1]).clone()
        deg[rows] += extra
        rp2 = torch.zeros(M + 1, dtype=torch.int64, device="cuda"); rp2[1:] = torch.cumsum(deg, 0)
        ci2 = torch.randint(0, K, (int(rp2[-1]),), generator=gen, device="cuda", dtype=torch.int32)
        # the original entries first in every row, the extra ones after them
        src_row = torch.repeat_interleave(torch.arange(M, device="cuda"), rp[1:] - rp[:-1])
        dst = rp2[src_row] + (torch.arange(ci.numel(), device="cuda") - rp[src_row])
        ci2[dst] = ci
        rp, ci = rp2, ci2
    rp = rp.to(torch.int32).contiguous(); ci = ci.contiguous()
    nnz = int(ci.numel())
    val = torch.rand(nnz, device="cuda") - 0.5
    for N in (128, 256):
        B = torch.rand(K, N, device="cuda") - 0.5
        C = torch.empty(M, N, device="cuda")
        res = {}
        for kern in ("auto", "stream", "seg-stream", "staged"):
            plan = spmm.SpmmPlan(rp, ci, K, N, values=val, kernel=kern)
            res[kern] = timed(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan))
            if kern == "auto": what = plan.describe().split("|")[-1].strip()[:95]; maxdeg = plan.describe().split("max_degree=")[1].split()[0]
            del plan
        print("hub rows %4d (longest row %6s, %d entries) N=%3d: auto %8.1f us | stream %8.1f  seg-stream %8.1f  staged %8.1f | %s"
              % (nhub, maxdeg, nnz, N, res["auto"], res["stream"], res["seg-stream"], res["staged"], what), flush=True)
