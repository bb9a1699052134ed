#!/usr/bin/env python3
"""The reference's run_test.sh protocol on the stand-ins: write each graph as .mtx, run the spmm_test driver on it (200 timed launches
per width, vendor rocSPARSE column beside the GE-SpMM column) through the plain call and through a plan; prints the CSV lines.
    python scripts/driver_compare.py [graph ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gespmm_amd  # noqa: E402,F401
from gespmm_amd import graphs  # noqa: E402

names = sys.argv[1:] or ["com-amazon-sbm", "com-amazon-like", "cit-hepth-like"]
drv = os.path.join(ROOT, "gespmm_amd", "lib", "spmm_test")
for name in names:
    g = graphs.synthetic_graph(name, seed=42, device="cuda")
    path = "/tmp/%s.mtx" % name
    graphs.write_mtx(path, g["rowptr"], g["colind"])
    for label, extra in (("plain call", []), ("plan (default life = the 200 timed launches)", ["--plan"]), ("plan + tune", ["--plan", "--tune"])):
        out = "/tmp/driver_%s.csv" % name
        if os.path.exists(out):
            os.remove(out)
        r = subprocess.run([drv, path, "0", "--out", out, "--seed", "1", "--method", "-1", "--use-values"] + extra, capture_output=True, text=True)
        line = open(out).read().strip() if os.path.exists(out) else "(no csv: %s)" % r.stderr.strip()[-300:]
        print("%-16s %-46s rocsparse/gespmm GFLOP/s at N=128,256,512: %s" % (name, label, line))
