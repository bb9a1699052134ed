#!/usr/bin/env python3
"""Padded-record kernel (spmm_records.hip) against the plan's AUTO choice at narrow widths: batches per task (the work a task is cut at); bits compared with the plain call.
    GESPMM_REC_BATCHES is read once per process: this script re-executes itself per value.
    python profiles/r06/scripts/records_sweep.py [graph ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))

if "GESPMM_REC_BATCHES" not in os.environ:
    for rows in os.environ.get("BATCHES", "2,3,4,6").split(","):
        env = dict(os.environ, GESPMM_REC_BATCHES=rows)
        subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, check=False)
    sys.exit(0)

import torch  # noqa: E402

from gespmm_amd import _lib, graphs, spmm  # noqa: E402
from kernel_ab import timeit  # noqa: E402

dev = torch.device("cuda")
rows = int(os.environ["GESPMM_REC_BATCHES"])
first = rows == int(os.environ.get("BATCHES", "2,3,4,6").split(",")[0])
widths = [int(x) for x in os.environ.get("WIDTHS", "32,16,64").split(",")]
for name in sys.argv[1:] or ["com-amazon-sbm", "com-amazon-like"]:
    if name.endswith("/4"):
        g = graphs.synthetic_graph(name[:-2], seed=42, device=dev, scale=0.25)
    elif os.path.exists(os.path.join(ROOT, "profiles", "r05", "holdout", name + ".npz")):
        import holdout_audit
        g = holdout_audit.from_npz(os.path.join(ROOT, "profiles", "r05", "holdout", name + ".npz"))
    else:
        g = graphs.synthetic_graph(name, seed=42, device=dev)
    M, K, nnz, rp, ci = g["M"], g["K"], g["nnz"], g["rowptr"], g["colind"]
    val = torch.rand(nnz, device=dev) - 0.5
    for N in widths:
        B = torch.rand(K, N, device=dev) - 0.5
        C = torch.empty((M, N), device=dev)
        alg = 4.0 * (M + 1) + 8.0 * nnz + 4.0 * (M + K) * N
        spmm.csr_spmm(rp, ci, val, B, out=C)
        ref = C.clone()
        if first:
            t = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C), 20)
            print("%s N=%d plain call %.1f us (%.3f)" % (name, N, t, alg / t / 8e6), flush=True)
            p = spmm.SpmmPlan(rp, ci, K, N, values=val, expected_launches=1000000)
            t = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p), 20)
            print("%s N=%d AUTO plan %.1f us (%.3f) | %s" % (name, N, t, alg / t / 8e6, p.describe().split("|")[-1].strip()[:90]), flush=True)
            del p
        for reorder in (True,):
            for fl in (0,):
                try:
                    p = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=reorder, kernel="records", flags=fl, expected_launches=1000000)
                except Exception as ex:  # noqa: BLE001
                    print("  records: %s" % str(ex)[:80])
                    continue
                C.zero_()
                t = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p), 50)
                ok = torch.equal(C.view(torch.int32), ref.view(torch.int32))
                d = p.describe()
                print("  %s N=%d records batches/task>=%-2d %-9s nt=%d: %.1f us (%.3f) fill %s tasks %s tables %s%s" % (
                    name, N, rows, "clustered" if reorder else "storage", 1 if fl else 0, t, alg / t / 8e6,
                    d.split("slot_fill=")[1].split(" ")[0], d.split("padded-records tasks=")[1].split(" ")[0], d.split("tables=")[1].split(" ")[0],
                    "" if ok else " BITS-DIFFER"), flush=True)
                del p
