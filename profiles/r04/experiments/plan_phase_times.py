"""Per-phase times of gespmm_plan_create (GESPMM_PLAN_TIMING=1 synchronises between phases) on a tiny and on the headline graph."""
import os, sys
os.environ["GESPMM_PLAN_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
import gespmm_amd  # noqa
from gespmm_amd import graphs, spmm
dev = torch.device("cuda")
for name in ("pubmed-selfloop-like", "com-amazon-sbm"):
    g = graphs.synthetic_graph(name, seed=42, device=dev)
    val = torch.rand(g["nnz"], device=dev) - 0.5
    for i in range(3):
        sys.stderr.write("==== %s creation %d\n" % (name, i))
        sys.stderr.flush()
        p = spmm.SpmmPlan(g["rowptr"], g["colind"], g["K"], 128, values=val)
        torch.cuda.synchronize()
        del p
