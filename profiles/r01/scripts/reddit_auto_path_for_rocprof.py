#!/usr/bin/env python3
"""reddit-like N=128 (config C2b): a few launches of the auto path, for rocprofv3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gespmm_amd
from gespmm_amd import _lib as F, graphs, spmm

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "reddit-like"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
g = graphs.synthetic_graph(name, device=dev)
M, K, nnz = g["M"], g["K"], g["nnz"]
val = torch.rand(nnz, device=dev)
B = torch.rand(K, N, device=dev); C = torch.empty(M, N, device=dev)
import json
cfg = json.loads(os.environ["GESPMM_CFG"]) if "GESPMM_CFG" in os.environ else None
plan = spmm.SpmmPlan(g["rowptr"], g["colind"], K, N) if os.environ.get("PLAN") == "1" else None
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
spmm.csr_spmm(g["rowptr"], g["colind"], val, B, out=C, cfg=cfg, plan=plan)
torch.cuda.synchronize(); e0.record()
for _ in range(iters):
    spmm.csr_spmm(g["rowptr"], g["colind"], val, B, out=C, cfg=cfg, plan=plan)
e1.record(); torch.cuda.synchronize()
print("%s N=%d: %.1f us per call" % (name, N, e0.elapsed_time(e1) / iters * 1e3))
