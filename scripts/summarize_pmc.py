#!/usr/bin/env python3
"""Condense rocprofv3 counter_collection CSVs to one line per (kernel, counter): mean over dispatches.
usage: summarize_pmc.py out.csv in1.csv [in2.csv ...]"""
import collections
import csv
import sys

out, ins = sys.argv[1], sys.argv[2:]
agg = collections.OrderedDict()
meta = {}
for f in ins:
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"], r["Counter_Name"])
        agg.setdefault(k, []).append(float(r["Counter_Value"]))
        meta[r["Kernel_Name"]] = (r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["LDS_Block_Size"])
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "grid", "workgroup", "vgpr", "lds_bytes", "counter", "dispatches", "mean_per_dispatch"])
    for (k, c), v in agg.items():
        w.writerow([k, *meta[k], c, len(v), "%.1f" % (sum(v) / len(v))])
