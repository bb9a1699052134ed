#!/usr/bin/env python3
"""C-store policy of the STREAMING kernels (plain / nt = 0x2 / sc1 = 0x8000) through plans and the plain call, after `sc1 nt` won 8 %
in the staged-rows kernel (staged_store_scope.log).   python profiles/r06/scripts/store_flags_sweep.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402

from gespmm_amd import graphs, spmm  # noqa: E402
from kernel_ab import timeit, load  # noqa: E402

dev = torch.device("cuda")
cases = [("com-amazon-sbm", (32, 64, 128)), ("com-amazon-like", (32, 128)), ("products-sbm", (32, 64)), ("reddit-sbm", (128,)), ("rmat-22", (256,)),
         ("powerlaw-ba", (128,))]
for name, widths in cases:
    if name == "powerlaw-ba":
        g = graphs.synthetic_graph("ba-m6", seed=42, device=dev)
    else:
        g = load(name, 1.0)
    M, K, nnz, rp, ci = g["M"], g["K"], g["nnz"], g["rowptr"], g["colind"]
    val = torch.rand(nnz, device=dev) - 0.5
    iters = 30 if nnz < 8e6 else (10 if nnz < 5e7 else 4)
    for N in widths:
        B = torch.rand(K, N, device=dev) - 0.5
        C = torch.empty((M, N), device=dev)
        spmm.csr_spmm(rp, ci, val, B, out=C)
        ref = C.clone()
        out = []
        for fl, tag in ((0, "plain-store"), (0x2, "nt"), (0x8000, "sc1")):
            t = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, cfg={"flags": fl}), iters)
            out.append("call/%s %.1f" % (tag, t))
        for kern in ("auto", "stream", "seg-stream"):
            for fl, tag in ((0, "plain-store"), (0x2, "nt"), (0x8000, "sc1")):
                try:
                    kw = {"expected_launches": 1000000} if kern == "auto" else {"reorder": "auto", "kernel": kern, "expected_launches": 1000000}
                    p = spmm.SpmmPlan(rp, ci, K, N, values=val, flags=fl, **kw)
                except Exception as ex:  # noqa: BLE001
                    out.append("%s/%s n/a" % (kern, tag))
                    continue
                C.zero_()
                t = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p), iters)
                ok = torch.equal(C.view(torch.int32), ref.view(torch.int32))
                d = p.describe().split("|")[-1].strip().split(" ")
                out.append("%s/%s %.1f%s%s" % (kern, tag, t, "" if ok else "(bits differ: long-row pass?)", " [" + d[0] + " " + d[1] + "]" if kern == "auto" and fl == 0 else ""))
                del p
        print("%-16s N=%-3d | %s" % (name, N, " | ".join(out)), flush=True)
        del B, C, ref
    del g, rp, ci, val
    torch.cuda.empty_cache()
