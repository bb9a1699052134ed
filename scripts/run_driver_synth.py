#!/usr/bin/env python3
"""run_test.sh for the synthetic stand-ins: write each graph as .mtx under /tmp, run the
spmm_test driver on it (vendor rocSPARSE column + GE-SpMM column) and print the CSV line."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gespmm_amd
from gespmm_amd import graphs

names = sys.argv[1:] or ["cit-hepth-like", "com-amazon-like"]
out = "/tmp/spmm_test_out.out"
if os.path.exists(out):
    os.remove(out)
with open(out, "a") as f:
    f.write("data,K=128-rocsparse-gflops,K=128-gespmm-gflops,K=256-rocsparse-gflops,K=256-gespmm-gflops,"
            "K=512-rocsparse-gflops,K=512-gespmm-gflops,\n")
for name in names:
    if name.endswith(".mtx"):
        path = name
    else:
        g = graphs.synthetic_graph(name, seed=42, device="cuda")
        path = "/tmp/%s.mtx" % name
        graphs.write_mtx(path, g["rowptr"], g["colind"])
    with open(out, "a") as f:
        f.write(os.path.basename(path) + ",")
    for method in ("-1",):
        r = subprocess.run([os.path.join(ROOT, "gespmm_amd", "lib", "spmm_test"), path, "0", "--out", out,
                            "--seed", "1", "--method", method], capture_output=True, text=True)
        print(r.stdout.strip()); print(r.stderr.strip()[-500:])
    with open(out, "a") as f:
        f.write("\n")
print(open(out).read())
