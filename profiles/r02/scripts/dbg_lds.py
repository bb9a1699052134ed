import sys; sys.path.insert(0,'.')
import numpy as np, torch, gespmm_amd
from gespmm_amd import graphs, spmm
g = graphs.load_mtx_as_csr("tests/golden/cora.mtx")
rp, ci = torch.from_numpy(g["rowptr"]).cuda(), torch.from_numpy(g["colind"]).cuda()
M, K = g["M"], g["K"]
val = torch.rand(g["nnz"], device="cuda") - 0.5
for N in (128, 32):
    B = torch.rand(K, N, device="cuda") - 0.5
    ref = spmm.csr_spmm(rp, ci, val, B)
    plan = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True, kernel="lds-rows")
    C = torch.full((M, N), float("nan"), device="cuda")
    spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan)
    bad = (C.view(torch.int32) != ref.view(torch.int32)).any(1).cpu().numpy()
    order = plan.order().numpy()
    pos = np.empty(M, np.int64); pos[order] = np.arange(M)
    badpos = np.sort(pos[bad])
    print("N", N, "bad rows", bad.sum(), "of", M, "nan rows", int(torch.isnan(C).any(1).sum()))
    print(" bad positions (first 40):", badpos[:40], " last:", badpos[-10:] if len(badpos) else None)
    deg = np.diff(g["rowptr"])
    print(" degrees of bad rows (first 20):", deg[bad][:20])
    print(plan.describe())
