#!/usr/bin/env python3
"""profiles/<round>/pmc_*/ (summary.csv + kernel_stats.csv from scripts/gpu_pmc.sh) -> one digest: per capture the product kernel(s),
fabric bytes, L2 hit rate, average duration, rate, average fabric read latency.    python scripts/pmc_digest.py profiles/r06"""
import csv
import glob
import os
import sys

root = sys.argv[1]
out = ["PMC captures under %s (scripts/gpu_profile_r06.sh + gpu_profile_r06b.sh; rocprofv3 --pmc in separate passes, summarize_pmc.py per (kernel, grid))." % root,
       "2*FETCH_SIZE + WRITE_SIZE in bytes (gfx950: FETCH_SIZE counts 128-byte fabric reads at 64; TCC_EA0_RDREQ x 128 B agrees to 0.01 %).",
       "Durations: rocprofv3 --kernel-trace --stats of the same command (one launch of the same kernel name by gespmm_init on its small built-in",
       "matrix, ~13 us, is taken out of the few-launch averages).", ""]
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    rows = list(csv.DictReader(open(d + "/summary.csv")))
    best = {}
    for r in rows:
        k = (r["kernel"], r["grid"])
        best[k] = max(best.get(k, 0), int(float(r["dispatches"])))
    top = max(best.values())
    ks = list(csv.reader(open(d + "/kernel_stats.csv")))
    out.append("== %s" % os.path.basename(d))
    for (kern, grid), n in best.items():
        if n < top or n < 3:
            continue
        c = {r["counter"]: float(r["mean_per_dispatch"]) for r in rows if r["kernel"] == kern and r["grid"] == grid}
        if "FETCH_SIZE" not in c:
            continue
        t = None
        for row in ks[1:]:
            if row and row[0] == kern:
                calls, total = int(row[1]), float(row[2])
                per = n // 2 if n >= 6 else n
                t = (total - 13000.0) / per / 1e3 if calls == per + 1 else total / calls / 1e3
        b = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        hit = c.get("TCC_HIT_sum", 0) / max(c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0), 1)
        lat = (c["TCC_EA0_RDREQ_LEVEL_sum"] / c["TCC_EA0_RDREQ_sum"]) if c.get("TCC_EA0_RDREQ_sum") and c.get("TCC_EA0_RDREQ_LEVEL_sum") else None
        out.append("  %-72s 2*FETCH+WRITE = %9.1f MB  L2 hits %.3f%s%s" % (
            kern.replace("void gespmm::", "").replace("(anonymous namespace)::", "")[:72], b / 1e6, hit,
            ("  %.1f us -> %.2f TB/s" % (t, b / t / 1e6)) if t else "", ("  read latency %.0f clk" % lat) if lat else ""))
open(os.path.join(root, "pmc_digest.log"), "w").write("\n".join(out) + "\n")
print("\n".join(out[5:]))
