#!/usr/bin/env python3
"""profiles/hbm_traffic.json from the rocprofv3 PMC summaries (scripts/gpu_pmc.sh): bytes per launch =
2 * FETCH_SIZE + WRITE_SIZE (KiB -> B; on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads,
MI355X_MICROARCH.md HBM section), L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS).
usage: update_traffic_json.py key=summary.csv [key=summary.csv ...]"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_fingerprint  # noqa: E402  (bench.py refuses to QUOTE an entry whose fingerprint is not the tree's)

path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
data = json.load(open(path)) if os.path.exists(path) else {}
for arg in sys.argv[1:]:
    key, f = arg.split("=", 1)
    # the measured kernel = the one with the most dispatches in the run (since round 6 gespmm_init launches every kernel family once
    # before the timed loop: one-dispatch rows that are not the product)
    rows = list(csv.DictReader(open(f)))
    disp = {}
    for r in rows:
        key_ = (r["kernel"], r["grid"])
        disp[key_] = max(disp.get(key_, 0), int(float(r["dispatches"])))
    kern, grid = max(disp, key=disp.get)
    c = {r["counter"]: float(r["mean_per_dispatch"]) for r in rows if r["kernel"] == kern and r["grid"] == grid}
    entry = {
        "FETCH_SIZE_KiB": c["FETCH_SIZE"], "WRITE_SIZE_KiB": c["WRITE_SIZE"],
        "bytes_per_launch": int(round((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)),
        "l2_hit_rate": round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4),
        "kernel": kern,
        "csrc_sha16": csrc_fingerprint(),
        "source": "%s (2*FETCH_SIZE + WRITE_SIZE, KiB->B; separate --pmc passes of `%s`; memory-side requests include Infinity-Cache hits)"
                  % (os.path.relpath(f, ROOT),
                     ("python scripts/kernel_pmc_case.py products-sbm %s auto 3" % key.split("/N")[1].split("/")[0]) if key.startswith("products-sbm")
                     else "python bench.py --no-extra --no-cpu-baseline --steps 50 --warmup 5 ..."),
    }
    data[key] = entry
    print(key, entry["bytes_per_launch"], entry["l2_hit_rate"])
json.dump(data, open(path, "w"), indent=1)
