"""Where the column-slab rule should start: planted-community graphs of the reddit stand-in's size (233 k rows) at mean degrees 64 ... 492
(communities of ~2.4 x the intra-community degree, two thirds of a row's entries inside), N = 128 — AUTO without slab tables
(GESPMM_SLABS=-1) against the slab tables asked for by name (round(mean / 64) ranges).    python profiles/r06/scripts/slab_density.py"""
import os, statistics, subprocess, sys
import torch
sys.path.insert(0, ".")

if len(sys.argv) > 1:
    import gespmm_amd
    from gespmm_amd import _lib, graphs, spmm
    d, kern = int(sys.argv[1]), sys.argv[2]
    M = 232965
    nnz = M * d
    nnz -= nnz % 2
    intra = 0.67 * d
    n_comm = max(4, int(M / (2.4 * intra)))
    rp, ci, _ = graphs.community_csr(M, nnz, n_comm, max(2, n_comm // 18), intra, 0.6, 1.5, 1.55, 42, "cuda")
    nnz = int(ci.numel())
    val = torch.rand(nnz, device="cuda") - 0.5
    B = torch.rand(M, 128, device="cuda") - 0.5
    C = torch.empty(M, 128, device="cuda")
    plan = spmm.SpmmPlan(rp, ci, M, 128, values=val, kernel=kern)
    fn = lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan)
    for _ in range(2): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev)
    ref = spmm.csr_spmm(rp, ci, val, B, cfg={"flags": _lib.FLAG_STRICT_ORDER})
    print("mean degree %3d (%d communities) GESPMM_SLABS=%s kernel=%-12s %8.1f us bits=%s | %s" % (
        d, n_comm, os.environ.get("GESPMM_SLABS", "-"), kern, t, bool(torch.equal(C.view(torch.int32), ref.view(torch.int32))),
        plan.describe().split("|")[-1].strip()[:120]), flush=True)
else:
    for d in (64, 96, 128, 160, 192, 256, 350, 492):
        for env, kern in (("-1", "auto"), (None, "staged-slabs")):
            e = dict(os.environ)
            if env: e["GESPMM_SLABS"] = env
            else: e.pop("GESPMM_SLABS", None)
            subprocess.run([sys.executable, __file__, str(d), kern], env=e, timeout=300)
