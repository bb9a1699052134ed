"""Analysis-stage timings: python scripts/plan_time.py [graph ...]  (device vs host analysis, warm and cold)"""
import sys, time
import torch
sys.path.insert(0, ".")
import gespmm_amd
from gespmm_amd import graphs, spmm

names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["com-amazon-like", "com-amazon-sbm", "products-sbm", "products-like"]
for name in names:
    g = graphs.synthetic_graph(name, seed=42, device="cuda")
    val = torch.rand(g["nnz"], device="cuda") - 0.5
    for analysis in ("device", "device", "device", "host"):
        if analysis == "host" and g["nnz"] > 5e7 and "--host-big" not in sys.argv:
            continue
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan = spmm.SpmmPlan(g["rowptr"], g["colind"], g["K"], 128, values=val, analysis=analysis)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        print("%-16s %-6s %8.2f ms | %s" % (name, analysis, dt, plan.describe()[:230]), flush=True)
        del plan
    del g, val
    torch.cuda.empty_cache()
