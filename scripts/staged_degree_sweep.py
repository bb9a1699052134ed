"""Where does the staged-rows kernel start to pay? Planted-community graphs of 600 k rows with mean degree 8 ... 48 (communities of ~48 rows,
intra-community degree = 2/3 of the mean), N = 128 and 256: staged-rows against the streaming kernels of the same plan, and what AUTO takes.
    python scripts/staged_degree_sweep.py"""
import statistics, sys
import torch
sys.path.insert(0, ".")  # run from the repository root
import gespmm_amd
from gespmm_amd import graphs, spmm


def timed(fn, reps=20):
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev)


M = 600_000
for mean in [int(x) for x in (sys.argv[1].split(',') if len(sys.argv) > 1 else '8,12,16,24,32,48'.split(','))]:
    nnz = M * mean
    rp, ci, _ = graphs.community_csr(M, nnz, M // 48, 512, mean * 2.0 / 3.0, 0.6, 1.5, 1.55, 42, "cuda")
    val = torch.rand(nnz, device="cuda") - 0.5
    for N in (128, 256):
        B = torch.rand(M, N, device="cuda") - 0.5
        C = torch.empty(M, N, device="cuda")
        t = {}
        for kern in ("auto", "stream", "seg-stream", "staged"):
            plan = spmm.SpmmPlan(rp, ci, M, N, values=val, kernel=kern)
            t[kern] = timed(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan))
            if kern == "auto": took = "staged-rows" if "kernel=staged-rows" in plan.describe() else "streaming"
            if kern == "staged": frac = plan.describe().split("staged_entries=")[1].split()[0] if "staged_entries=" in plan.describe() else "-"
            del plan
        best_stream = min(t["stream"], t["seg-stream"])
        verdict = "" if (took == "staged-rows") == (t["staged"] < best_stream) or abs(t["staged"] / best_stream - 1) < 0.03 else "   <-- AUTO picked the slower one"
        print("mean degree %2d N=%3d: staged %8.1f us  streaming %8.1f us (batch %.1f, segmented %.1f)  x%.2f  staged share %s | AUTO took %s (%.1f us)%s"
              % (mean, N, t["staged"], best_stream, t["stream"], t["seg-stream"], best_stream / t["staged"], frac, took, t["auto"], verdict), flush=True)
    del rp, ci, val
