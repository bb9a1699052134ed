#!/usr/bin/env python3
"""AUTO plans against the plain call on a grid of graphs and widths: does the analysis stage ever lose?
    python scripts/plan_audit.py [--baseline profiles/r03/plan_audit.log]

With --baseline every (graph, N) is also compared with the same line of an earlier log: the measure is plan / plain of the SAME
run (boxes differ by a few per cent in absolute time, the plain call is the yardstick that travels); a line whose ratio is more
than 6 % worse than the baseline's AND whose plan time is more than 4 % worse is marked REGRESSION and the script exits 1 — logs
from different boxes differ by +-4 %; the sharp check is scripts/plan_regression_ab.py (the previous round's library in the same process) (the structureless stand-in's plan
lost 5 % between rounds 2 and 3 and nobody noticed because the headline had moved to another graph)."""
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import gespmm_amd  # noqa: F401,E402
from gespmm_amd import graphs, spmm  # noqa: E402

dev = torch.device("cuda")

BASE = {}
if "--baseline" in sys.argv:
    with open(sys.argv[sys.argv.index("--baseline") + 1]) as fh:
        for ln in fh:
            m = re.match(r"(.+?)\s+N=(\d+)\s+plain\s+([0-9.]+) us\s+plan\s+([0-9.]+) us", ln)
            if m:
                BASE[(m.group(1).strip(), int(m.group(2)))] = (float(m.group(4)) / float(m.group(3)), float(m.group(4)))
REGRESSIONS = []


def timeit(fn, iters):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def cases():
    yield "com-amazon-like", graphs.synthetic_graph("com-amazon-like", seed=42, device=dev)
    yield "com-amazon-like locality 0.9", graphs.synthetic_graph("com-amazon-like", seed=42, device=dev, locality=0.9)
    yield "com-amazon-sbm", graphs.synthetic_graph("com-amazon-sbm", seed=42, device=dev)
    yield "com-amazon-sbm (planted order, not shuffled)", None
    yield "cit-hepth-like", graphs.synthetic_graph("cit-hepth-like", seed=42, device=dev)
    for name in ("cora", "pubmed"):
        g = graphs.load_mtx_as_csr(os.path.join(ROOT, "tests", "golden", name + ".mtx"))
        yield name, {"M": g["M"], "K": g["K"], "nnz": g["nnz"], "rowptr": torch.from_numpy(g["rowptr"]).to(dev),
                     "colind": torch.from_numpy(g["colind"]).to(dev)}
    for s in (16, 18, 20):
        yield "rmat-%d" % s, graphs.rmat_shard(s, 16, 0, 1, seed=42, device=dev)
    yield "products-sbm x0.25", graphs.synthetic_graph("products-sbm", seed=42, device=dev, scale=0.25)
    yield "products-like x0.25", graphs.synthetic_graph("products-like", seed=42, device=dev, scale=0.25)
    yield "reddit-like x0.1", graphs.synthetic_graph("reddit-like", seed=42, device=dev, scale=0.1)


for name, g in cases():
    if g is None:
        M, nnz = graphs.SPECS["com-amazon-like"][:2]
        rp, ci, _ = graphs.community_csr(M, nnz, 75149, 1024, 5.0, 0.6, 1.5, 1.55, 42, dev, shuffle=False)
        g = {"M": M, "K": M, "nnz": int(ci.numel()), "rowptr": rp, "colind": ci}
    M, K, nnz = g["M"], g["K"], g["nnz"]
    rp, ci = g["rowptr"], g["colind"]
    val = torch.rand(nnz, device=dev) - 0.5
    for N in (32, 128, 256, 512):
        if 4.0 * (M + K) * N > 40e9:
            continue
        B = torch.rand(K, N, device=dev) - 0.5
        C = torch.empty((M, N), device=dev)
        iters = 50 if nnz < 5e6 else 10
        t_plain = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C), iters)
        ref = C.clone()
        t0 = time.time()
        plan = spmm.SpmmPlan(rp, ci, K, N, values=val)
        dt = time.time() - t0
        t_plan = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan), iters)
        same = torch.equal(C.view(torch.int32), ref.view(torch.int32))
        flag = "  <-- plan loses" if t_plan > 1.05 * t_plain else ""
        kern = "staged-rows" if "kernel=staged-rows" in plan.describe() else ""
        if kern:  # AUTO took the staged-rows kernel: what would the streaming kernels of the same plan have done?
            best = None
            for k2 in ("stream", "seg-stream"):
                p2 = spmm.SpmmPlan(rp, ci, K, N, values=val, kernel=k2)
                t2 = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p2), iters)
                best = t2 if best is None else min(best, t2)
                del p2
            kern += " (streaming kernels of the same plan: %.1f us, x%.2f)" % (best, best / t_plan)
            if t_plan > 1.03 * best:
                flag += "  <-- staged loses"
        prev = BASE.get((name, N))
        # both measures must be worse: the ratio (robust against a slower box) and the plan's own time (robust against a noisy plain
        # figure in either log); small graphs are launch-latency noise
        if prev is not None and nnz >= (1 << 20) and t_plan / t_plain > 1.06 * prev[0] and t_plan > 1.04 * prev[1]:
            flag += "  <-- REGRESSION: plan/plain %.3f (%.1f us), baseline %.3f (%.1f us)" % (t_plan / t_plain, t_plan, prev[0], prev[1])
            REGRESSIONS.append((name, N))
        print("%-46s N=%-3d plain %9.1f us  plan %9.1f us  x%.2f  bits=%s  analysis %.2fs  %s %s%s" %
              (name, N, t_plain, t_plan, t_plain / t_plan, "same" if same else "LONG-ROW-REASSOC", dt,
               plan.describe().split(" ")[0], kern, flag), flush=True)
        del plan, B, C, ref
    del g
    torch.cuda.empty_cache()
if BASE:
    print("# against the baseline log: %d line(s) compared, %d regression(s) %s" % (len(BASE), len(REGRESSIONS), REGRESSIONS or ""))
    sys.exit(1 if REGRESSIONS else 0)
