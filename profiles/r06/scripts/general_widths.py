#!/usr/bin/env python3
"""Round 6: the plan's fast kernel at feature widths that are not powers of two, and the max reducer, on the headline graph (and the
products-shaped community graph at 1/4 size): plain call | AUTO plan | forced staged plan | best streaming kernel of the same plan.
    python profiles/r06/scripts/general_widths.py [graph ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402

from gespmm_amd import graphs, spmm  # noqa: E402
from kernel_ab import timeit  # noqa: E402

dev = torch.device("cuda")
names = sys.argv[1:] or ["com-amazon-sbm", "products-sbm@0.25"]
for name in names:
    nm, _, sc = name.partition("@")
    g = graphs.synthetic_graph(nm, seed=42, device=dev, **({"scale": float(sc)} if sc else {}))
    M, K, nnz, rp, ci = g["M"], g["K"], g["nnz"], g["rowptr"], g["colind"]
    val = torch.rand(nnz, device=dev) - 0.5
    iters = 30 if nnz < 8e6 else 6
    print("%s M=%d nnz=%d" % (name, M, nnz), flush=True)
    for N in (41, 47, 64, 100, 128, 200, 256, 602):
        B = torch.rand(K, N, device=dev) - 0.5
        C = torch.empty((M, N), device=dev)
        alg = 4.0 * (M + 1) + 8.0 * nnz + 4.0 * (M + K) * N
        spmm.csr_spmm(rp, ci, val, B, out=C)
        ref = C.clone()
        out = ["plain %.1f" % timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C), iters)]
        for kern in ("auto", "staged", "stream", "seg-stream"):
            kw = {"expected_launches": 1000000} if kern == "auto" else {"reorder": True, "kernel": kern}
            p = spmm.SpmmPlan(rp, ci, K, N, values=val, **kw)
            C.zero_()
            t = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p), iters)
            ok = torch.equal(C.view(torch.int32), ref.view(torch.int32))
            d = p.describe()
            tag = ""
            if "kernel=staged-rows" in d:
                tag = " [staged %s]" % d.split("staged_entries=")[1].split(" ")[0]
            out.append("%s %.1f (%.3f)%s%s" % (kern, t, alg / (t * 1e-6) / 8e12, tag, "" if ok else " BITS-DIFFER"))
            del p
        print("  N=%-4d %s" % (N, " | ".join(out)), flush=True)
        if N in (128, 200):
            want = spmm.csr_spmm_max(rp, ci, B)
            tp = timeit(lambda: spmm.csr_spmm_max(rp, ci, B), iters)
            p = spmm.SpmmPlan(rp, ci, K, N, expected_launches=1000000)
            got = p.run(None, B, reduce_max=-10000.0)
            tm = timeit(lambda: p.run(None, B, out=C, reduce_max=-10000.0), iters)
            print("         max reducer: plain %.1f | AUTO plan %.1f %s%s" % (tp, tm, "[staged]" if "kernel=staged-rows" in p.describe() else "[streaming]",
                                                                          "" if torch.equal(got.view(torch.int32), want.view(torch.int32)) else " BITS-DIFFER"), flush=True)
            del p
        del B, C, ref
    del g, rp, ci, val
    torch.cuda.empty_cache()
