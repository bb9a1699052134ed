#!/usr/bin/env python3
"""Where does the padded-record kernel win? AUTO plan (before the rule existed: streaming / lane-group staged kernels) against
kernel="records" on the clustered order, at N = 16 / 32 / 64, on the stand-ins and the hold-out graphs; bits compared with the plain call.
    python profiles/r06/scripts/records_audit.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402

from gespmm_amd import graphs, spmm  # noqa: E402
import holdout_audit  # noqa: E402
from kernel_ab import timeit  # noqa: E402

dev = torch.device("cuda")
HOLD = holdout_audit.HOLD


def cases():
    for n in ("com-amazon-sbm", "com-amazon-like", "cit-hepth-like"):
        yield n, (lambda n=n: graphs.synthetic_graph(n, seed=42, device=dev))
    yield "products-sbm/4", (lambda: graphs.synthetic_graph("products-sbm", seed=42, device=dev, scale=0.25))
    for n in sorted(f[:-4] for f in os.listdir(HOLD) if f.endswith(".npz")) if os.path.isdir(HOLD) else []:
        yield n, (lambda n=n: holdout_audit.from_npz(os.path.join(HOLD, n + ".npz")))
    yield "rmat-18", (lambda: graphs.rmat_shard(18, device=dev))


widths = [int(x) for x in os.environ.get("WIDTHS", "16,32,64").split(",")]
for name, make in cases():
    try:
        g = make()
    except Exception as ex:  # noqa: BLE001
        print("== %s skipped: %s" % (name, str(ex)[:80]), flush=True)
        continue
    M, K, nnz, rp, ci = g["M"], g["K"], g["nnz"], g["rowptr"], g["colind"]
    val = torch.rand(nnz, device=dev) - 0.5
    deg = rp[1:] - rp[:-1]
    print("== %s: M=%d nnz=%d mean degree %.1f max %d" % (name, M, nnz, nnz / M, int(deg.max())), flush=True)
    for N in widths:
        B = torch.rand(K, N, device=dev) - 0.5
        C = torch.empty((M, N), device=dev)
        iters = 30 if nnz < 2e7 else 8
        t_plain = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C), iters)
        ref = C.clone()
        row = "  N=%-3d plain %8.1f" % (N, t_plain)
        for label, kw in (("AUTO", {}), ("stream", {"reorder": True, "kernel": "stream"}), ("staged", {"reorder": True, "kernel": "staged"}),
                          ("records", {"reorder": True, "kernel": "records"}), ("records/storage", {"reorder": False, "kernel": "records"})):
            try:
                p = spmm.SpmmPlan(rp, ci, K, N, values=val, expected_launches=1000000, **kw)
            except Exception as ex:  # noqa: BLE001
                row += "  %s n/a(%s)" % (label, str(ex)[:20])
                continue
            d = p.describe()
            if label == "staged" and "kernel=staged-rows" not in d:
                del p
                continue
            if label.startswith("records") and "padded-records" not in d:
                row += "  %s not-built" % label
                del p
                continue
            C.zero_()
            t = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p), iters)
            ok = torch.equal(C.view(torch.int32), ref.view(torch.int32))
            extra = ""
            if label == "AUTO":
                extra = "[%s %s %s]" % (d.split(" ")[0], ("model " + d.split("l2_model=")[1].split(" ")[0]) if "l2_model=" in d else "",
                                        d.split("|")[-1].strip().split(" ")[0])
            if label == "records":
                extra = "[fill %s]" % d.split("slot_fill=")[1].split(" ")[0]
            row += "  %s %.1f%s%s" % (label, t, extra, "" if ok else " BITS-DIFFER")
            del p
        print(row, flush=True)
    del g, rp, ci, val
    torch.cuda.empty_cache()
