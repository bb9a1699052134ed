#!/usr/bin/env python3
"""Reference CPU loop beside the GPU number for the BASELINE.json configs:
the driver's --validate --cpu-baseline path (1 thread, the reference's i->k->ptr loop inside the driver
binary) on the small configs in full; bench.py reports the same for the headline workload."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gespmm_amd
from gespmm_amd import graphs

drv = os.path.join(ROOT, "gespmm_amd", "lib", "spmm_test")
for name, ncols in (("cit-hepth-like", "32"), ("cit-hepth-like", "128"), ("pubmed", "128"), ("com-amazon-like", "128")):
    if name == "pubmed":
        path = os.path.join(ROOT, "tests", "golden", "pubmed.mtx")
    else:
        g = graphs.synthetic_graph(name, seed=42, device="cuda")
        path = "/tmp/%s.mtx" % name
        if not os.path.exists(path):
            graphs.write_mtx(path, g["rowptr"], g["colind"])
    r = subprocess.run([drv, path, "0", "--ncols", ncols, "--validate", "--cpu-baseline", "--seed", "1", "--out", "/tmp/cpu_base.out"],
                       capture_output=True, text=True)
    print("== %s N=%s (host cores: %d)" % (name, ncols, os.cpu_count()))
    print("\n".join(l for l in r.stdout.splitlines() if "GFLOP" in l or "validat" in l.lower() or "error" in l.lower()))
    sys.stdout.flush()
