#!/usr/bin/env python3
"""Condense rocprofv3 counter_collection CSVs to one line per (kernel, grid, counter): mean over dispatches. (Per grid since round 6:
gespmm_init launches every kernel family once on a small built-in matrix before a run's own launches — the same kernel NAME, another
grid; averaged together, a run of three products read a quarter low.)
usage: summarize_pmc.py out.csv in1.csv [in2.csv ...]"""
import collections
import csv
import sys

out, ins = sys.argv[1], sys.argv[2:]
agg = collections.OrderedDict()
meta = {}
for f in ins:
    for r in csv.DictReader(open(f)):
        kg = (r["Kernel_Name"], r["Grid_Size"])
        agg.setdefault((kg, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        meta[kg] = (r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["LDS_Block_Size"])
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "grid", "workgroup", "vgpr", "lds_bytes", "counter", "dispatches", "mean_per_dispatch"])
    for (kg, c), v in agg.items():
        w.writerow([kg[0], *meta[kg], c, len(v), "%.1f" % (sum(v) / len(v))])
