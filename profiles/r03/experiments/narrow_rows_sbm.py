"""products-sbm through the plan at one width (for rocprofv3 --pmc): python narrow_rows_sbm.py N [plan kernel]"""
import sys, torch
sys.path.insert(0, ".")
import gespmm_amd
from gespmm_amd import graphs, spmm
N = int(sys.argv[1])
g = graphs.synthetic_graph("products-sbm", seed=42, device="cuda")
val = torch.rand(g["nnz"], device="cuda") - 0.5
B = torch.rand(g["K"], N, device="cuda") - 0.5
C = torch.empty(g["M"], N, device="cuda")
plan = spmm.SpmmPlan(g["rowptr"], g["colind"], g["K"], N, values=val, kernel=sys.argv[2] if len(sys.argv) > 2 else "auto")
print(plan.describe())
for _ in range(6): spmm.csr_spmm(g["rowptr"], g["colind"], val, B, out=C, plan=plan)
torch.cuda.synchronize()
