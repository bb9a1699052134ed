"""Headline graph at N = 32 (frac 0.29): which geometry? variant (V = 1 / 2 / 4 floats per lane => 32 / 16 / 8 lanes
per row => 2 / 4 / 8 rows per gather instruction) x streaming kernel x task size, through clustered plans. Same bits everywhere."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch

import gespmm_amd  # noqa
from gespmm_amd import graphs, spmm, _lib

dev = torch.device("cuda")


def med(fn, n=100):
    for _ in range(10):
        fn()
    s = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    e = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    torch.cuda.synchronize()
    for i in range(n):
        s[i].record()
        fn()
        e[i].record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) * 1e3 for a, b in zip(s, e))


for gname in sys.argv[1:] or ["com-amazon-sbm", "com-amazon-like"]:
    g = graphs.synthetic_graph(gname, seed=42, device=dev)
    M, K, nnz = g["M"], g["K"], g["nnz"]
    rp, ci = g["rowptr"], g["colind"]
    val = torch.rand(nnz, device=dev) - 0.5
    for N in (32, 16, 64):
        B = torch.rand(K, N, device=dev) - 0.5
        C = torch.empty(M, N, device=dev)
        ab = 4 * (M + 1) + 8 * nnz + 4 * K * N + 4 * M * N
        ref = spmm.csr_spmm(rp, ci, val, B)
        t_plain = med(lambda: spmm.csr_spmm(rp, ci, val, B, out=C))
        p0 = spmm.SpmmPlan(rp, ci, K, N, values=val)
        t_auto = med(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p0))
        print("== %s N=%d: plain %.1f us, AUTO plan %.1f us (frac %.3f) | %s" % (gname, N, t_plain, t_auto, ab / t_auto / 8e6, p0.describe()[-90:]), flush=True)
        for variant in (1, 2, 3):
            if variant == 3 and N % 4:
                continue
            for kern in ("stream", "seg-stream"):
                row = []
                for te in (0, 32, 64, 128, 256):
                    for flags in (0, _lib.FLAG_SHALLOW_UNROLL):
                        try:
                            p = spmm.SpmmPlan(rp, ci, K, N, variant=variant, values=val, kernel=kern, task_entries=te, flags=flags)
                        except Exception as ex:  # noqa: BLE001
                            row.append("te=%d err" % te)
                            continue
                        t = med(lambda: spmm.csr_spmm(rp, ci, val, B, variant=variant, out=C, plan=p), 50)
                        ok = torch.equal(C.view(torch.int32), ref.view(torch.int32))
                        row.append("te=%d%s %.1f%s" % (te, "/U4" if flags else "", t, "" if ok else " BITS!"))
                        del p
                print("   variant %d %-10s %s" % (variant, kern, "  ".join(row)), flush=True)
