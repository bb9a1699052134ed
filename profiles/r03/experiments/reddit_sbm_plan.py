"""The reddit-sized planted-community graph through its AUTO plan at one width (for rocprofv3 --pmc): python reddit_sbm_plan.py N"""
import sys, torch
sys.path.insert(0, ".")
import gespmm_amd
from gespmm_amd import graphs, spmm
N = int(sys.argv[1])
M, nnz = graphs.SPECS["reddit-like"][:2]
rp, ci, _ = graphs.community_csr(M, nnz, 290, 16, 330.0, 0.6, 1.5, 1.55, 42, "cuda")
val = torch.rand(int(ci.numel()), device="cuda") - 0.5
B = torch.rand(M, N, device="cuda") - 0.5
C = torch.empty(M, N, device="cuda")
plan = spmm.SpmmPlan(rp, ci, M, N, values=val)
print(plan.describe())
for _ in range(6): spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan)
torch.cuda.synchronize()
