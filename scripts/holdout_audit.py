#!/usr/bin/env python3
"""AUTO plan against every explicit choice on graphs the heuristics were NOT tuned on (scripts/holdout_graphs.py: networkx LFR,
Holme-Kim, Newman-Watts-Strogatz, Barabasi-Albert, random geometric; plus RMAT with non-Graph500 parameters).

    python scripts/holdout_audit.py [--only NAME ...] [--widths 32 128 256]
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum ... -- python scripts/holdout_audit.py --pmc-mode     (3 AUTO launches per case)

Per (graph, N): plain call, AUTO plan, and every explicit (order, kernel) pair; a line is flagged when AUTO is more than 5 %
behind the best explicit choice. Bits are compared with the plain call."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import gespmm_amd  # noqa: F401,E402
from gespmm_amd import graphs, spmm  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_ab  # noqa: E402

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


STANDINS = ("com-amazon-sbm", "com-amazon-like", "products-sbm", "products-like", "reddit-sbm", "reddit-like")


def cases(only, standins=False):
    if standins:  # the repository's own generators, same table (for comparison with the hold-out rows)
        for n in STANDINS:
            if not only or n in only:
                yield n, (lambda n=n: graphs.synthetic_graph(n, seed=42, device=dev))
        return
    names = sorted(f[:-4] for f in os.listdir(HOLD) if f.endswith(".npz")) if os.path.isdir(HOLD) else []
    for n in names:
        if not only or n in only:
            yield n, (lambda n=n: from_npz(os.path.join(HOLD, n + ".npz")))
    # Kronecker graphs away from the Graph500 parameters (.57, .19, .19, .05): flatter, and strongly skewed
    for tag, probs, scale, ef in (("rmat-flat(.45,.22,.22,.11) s20 ef8", (0.45, 0.22, 0.22, 0.11), 20, 8),
                                  ("rmat-skew(.65,.15,.15,.05) s19 ef24", (0.65, 0.15, 0.15, 0.05), 19, 24)):
        if not only or tag.split("(")[0] in only:
            yield tag, (lambda probs=probs, scale=scale, ef=ef: graphs.rmat_shard(scale, ef, 0, 1, seed=7, device=dev, probs=probs))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=[])
    ap.add_argument("--widths", nargs="*", type=int, default=[32, 128, 256])
    ap.add_argument("--pmc-mode", action="store_true")
    ap.add_argument("--standins", action="store_true", help="the repository's own stand-ins instead of the hold-out graphs")
    args = ap.parse_args()
    worst = 1.0
    for name, make in cases(args.only, args.standins):
        try:
            g = make()
        except Exception as ex:  # noqa: BLE001 - a missing / half-written file must not end the audit
            print("== %s: skipped (%s: %s)" % (name, type(ex).__name__, str(ex)[:80]), flush=True)
            continue
        M, K, nnz = g["M"], g["K"], g["nnz"]
        rp, ci = g["rowptr"], g["colind"]
        val = torch.rand(nnz, device=dev) - 0.5
        deg = (rp[1:] - rp[:-1])
        print("== %s: M=%d nnz=%d mean degree %.1f max %d" % (name, M, nnz, nnz / M, int(deg.max())), flush=True)
        for N in args.widths:
            B = torch.rand(K, N, device=dev) - 0.5
            C = torch.empty((M, N), device=dev)
            if 4.0 * (M + K) * N > 60e9:
                continue
            iters = 30 if nnz < 8e6 else (10 if nnz < 5e7 else 4)
            if args.pmc_mode:
                plan = spmm.SpmmPlan(rp, ci, K, N, values=val, expected_launches=1000000)  # steady state: the audit is about the kernel rules
                for _ in range(3):
                    spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan)
                torch.cuda.synchronize()
                print("PMC %s N=%d | %s" % (name, N, plan.describe()), flush=True)
                continue
            t_plain = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C), iters)
            ref = C.clone()
            t0 = time.time()
            plan = spmm.SpmmPlan(rp, ci, K, N, values=val, expected_launches=1000000)  # steady state: the audit is about the kernel rules
            dt = time.time() - t0
            t_auto = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan), iters)
            same = torch.equal(C.view(torch.int32), ref.view(torch.int32))
            desc = plan.describe()
            # the same plan after gespmm_plan_tune (kernel choice by measurement)
            plan.tune(B, out=C, reps=3)
            t_tuned = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan), iters)
            same = same and torch.equal(C.view(torch.int32), ref.view(torch.int32))
            del plan
            alts = {}
            for order in (True, False):
                for kern in ("stream", "seg-stream", "staged"):
                    if kern == "staged" and (N < 32 or not order):  # (round 6: the staged-rows kernel serves every width)
                        continue
                    try:
                        p2 = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=order, kernel=kern, expected_launches=1000000)
                    except Exception as ex:  # noqa: BLE001
                        alts["%s/%s" % ("clustered" if order else "storage", kern)] = (None, str(ex)[:40])
                        continue
                    d2 = p2.describe()
                    if kern == "staged" and "kernel=staged-rows" not in d2:
                        del p2
                        continue
                    t2 = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p2), iters)
                    note = kernel_ab.bits_note(C, ref, rp, ci, val, B).strip()
                    if kern == "staged":
                        note += "(share %s)" % d2.split("staged_entries=")[1].split(" ")[0]
                    alts["%s/%s" % ("clustered" if order else "storage", kern)] = (t2, note)
                    del p2
            if nnz / M > 96:  # dense graphs: the clustered order is reachable only without the cache-blocked path
                from gespmm_amd import _lib
                for kern in ("stream", "seg-stream"):
                    p2 = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True, kernel=kern, flags=_lib.FLAG_NO_SLAB_BLOCKED, expected_launches=1000000)
                    t2 = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p2), iters)
                    ok = torch.equal(C.view(torch.int32), ref.view(torch.int32))
                    alts["clustered-noslab/%s" % kern] = (t2, kernel_ab.bits_note(C, ref, rp, ci, val, B).strip() + " (model %s)" % p2.describe().split("l2_model=")[1].split(" ")[0])
                    del p2
            best_k, best_t = min(((k, v[0]) for k, v in alts.items() if v[0] is not None), key=lambda kv: kv[1])
            best_t = min(best_t, t_plain)
            ratio = t_auto / best_t
            worst = max(worst, ratio)
            flag = "  <-- AUTO %.0f %% behind %s" % (100 * (ratio - 1), best_k if best_t < t_plain else "plain") if ratio > 1.05 else ""
            m = desc.split("l2_model=")[1].split(" ")[0] if "l2_model=" in desc else "-"
            kern = desc.split("|")[-1].strip().split(" (")[0][:70]
            print("  N=%-3d plain %9.1f  AUTO %9.1f  AUTO+tune %9.1f us (x%.2f, bits %s, analysis %.3fs, %s, l2_model %s, %s)%s" %
                  (N, t_plain, t_auto, t_tuned, t_plain / t_auto, "same" if same else "REASSOC", dt, desc.split(" ")[0], m, kern, flag),
                  flush=True)
            print("        " + "  ".join("%s %s%s" % (k, ("%.1f" % v[0]) if v[0] is not None else "n/a", (" " + v[1]) if v[1] else "")
                                         for k, v in alts.items()), flush=True)
            del B, C, ref
        del g, rp, ci, val
        torch.cuda.empty_cache()
    if not args.pmc_mode:
        print("worst AUTO / best explicit: x%.3f" % worst)


if __name__ == "__main__":
    main()
