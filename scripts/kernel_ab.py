#!/usr/bin/env python3
"""Explicit plan kernels side by side on named graphs (round 5's kernel work; leaner than holdout_audit.py: one timing per choice).

    python scripts/kernel_ab.py --graphs geometric nws-k10 com-amazon-sbm --widths 128 --kernels stream staged [--auto]

Graph names: the repository's stand-ins (gespmm_amd/graphs.py) or <name>.npz under $GESPMM_HOLDOUT_DIR (default profiles/r05/holdout;
written by scripts/holdout_graphs.py). Bits are compared with the plain call. Env knobs of the library (GESPMM_STAGED_U ...) are read once per
process: run the script once per setting."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import gespmm_amd  # noqa: F401,E402
from gespmm_amd import graphs, spmm  # noqa: E402

dev = torch.device("cuda")
HOLD = os.environ.get("GESPMM_HOLDOUT_DIR", os.path.join(ROOT, "profiles", "r05", "holdout"))


def timeit(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def bits_note(C, ref, rp, ci, val, B):
    """"" when the bits are equal; a tolerance-class label when they differ only as a re-association of the sums may (the long-row pass
    on rows beyond max(2048, 32 x mean degree): |delta| <= 1e-4 * sum |a.b|); BITS-DIFFER otherwise."""
    if torch.equal(C.view(torch.int32), ref.view(torch.int32)):
        return ""
    bound = spmm.csr_spmm(rp, ci, val.abs(), B.abs()) if val is not None else spmm.csr_spmm_no_edge_value(rp, ci, B.abs())
    d = (C.double() - ref.double()).abs()
    if bool((d <= 1e-4 * bound.double() + 1e-30).all()):
        rows = int((d.amax(1) > 0).sum())
        return " tolerance-class(%d rows re-associated by the long-row pass, max |delta| / sum|a.b| %.1e)" % (rows, float((d / (bound.double() + 1e-30)).max()))
    return " BITS-DIFFER"


def from_npz(path):
    z = np.load(path)
    n = int(z["n"])
    lo, hi = torch.from_numpy(z["lo"]).to(dev).long(), torch.from_numpy(z["hi"]).to(dev).long()
    r, c = torch.cat([lo, hi]), torch.cat([hi, lo])
    order = torch.argsort(r * n + c)
    r, c = r[order], c[order]
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(torch.bincount(r, minlength=n), 0)
    return {"M": n, "K": n, "nnz": int(c.numel()), "rowptr": rowptr.to(torch.int32), "colind": c.to(torch.int32)}


def load(name, scale):
    p = os.path.join(HOLD, name + ".npz")
    if os.path.exists(p):
        return from_npz(p)
    if name.startswith("rmat-"):  # rmat-<scale>: the Graph500 generator of config 5, one shard = the whole graph
        return graphs.rmat_shard(int(name.split("-")[1]), device=dev)
    kw = {"scale": scale} if scale != 1.0 else {}
    return graphs.synthetic_graph(name, seed=42, device=dev, **kw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", nargs="+", required=True)
    ap.add_argument("--widths", nargs="*", type=int, default=[128])
    ap.add_argument("--kernels", nargs="*", default=["stream", "staged"])
    ap.add_argument("--auto", action="store_true", help="also the AUTO plan and the plain call")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    for name in args.graphs:
        g = load(name, args.scale)
        M, K, nnz, rp, ci = g["M"], g["K"], g["nnz"], g["rowptr"], g["colind"]
        val = torch.rand(nnz, device=dev) - 0.5
        iters = 30 if nnz < 8e6 else (10 if nnz < 5e7 else 4)
        for N in args.widths:
            B = torch.rand(K, N, device=dev) - 0.5
            C = torch.empty((M, N), device=dev)
            spmm.csr_spmm(rp, ci, val, B, out=C)
            ref = C.clone()
            alg = 4.0 * (M + 1) + 8.0 * nnz + 4.0 * (M + K) * N
            out = []
            if args.auto:
                t = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C), iters)
                out.append("plain %.1f" % t)
            for kern in (["auto"] if args.auto else []) + args.kernels:
                try:
                    p = spmm.SpmmPlan(rp, ci, K, N, values=val, **({"expected_launches": 1000000} if kern == "auto" else {"reorder": True, "kernel": kern}))
                except Exception as ex:  # noqa: BLE001
                    out.append("%s n/a (%s)" % (kern, str(ex)[:40]))
                    continue
                d = p.describe()
                C.zero_()
                t = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p), iters)
                okn = bits_note(C, ref, rp, ci, val, B)
                note = ""
                if "kernel=staged-rows" in d:
                    note = " share=" + d.split("staged_entries=")[1].split(" ")[0]
                elif kern == "staged":
                    note = " (not staged)"
                if kern == "auto":
                    note += " [" + d.split("|")[-1].strip().split(" ")[0] + "]"
                out.append("%s %.1f us frac %.3f%s%s" % (kern, t, alg / (t * 1e-6) / 8e12, note, okn))
                del p
            print("%s%-16s N=%-3d nnz=%d | %s" % (args.tag, name, N, nnz, " | ".join(out)), flush=True)
            del B, C, ref
        del g, rp, ci, val
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
