import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scripts"))
import torch
from gespmm_amd import graphs, spmm, _lib
from kernel_ab import timeit
g = graphs.load_mtx_as_csr("tests/golden/pubmed.mtx")
rp = torch.from_numpy(g["rowptr"]).cuda(); ci = torch.from_numpy(g["colind"]).cuda()
K = g["K"]; nnz = g["nnz"]
val = torch.rand(nnz, device="cuda")
for N in (128, 3):
    B = torch.rand(K, N, device="cuda"); C = torch.empty(g["M"], N, device="cuda")
    t0 = time.perf_counter(); p = spmm.SpmmPlan(rp, ci, K, N); torch.cuda.synchronize(); t_plan = (time.perf_counter() - t0) * 1e3
    print("N=%d plan made in %.2f ms: %s" % (N, t_plan, p.describe()[:200]))
    def wall(fn, n=300):
        for _ in range(10): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    print("   wall per call: plain %.1f us | plan (values passed each call) %.1f us | plan unweighted %.1f us | GPU-time plain %.1f plan %.1f" % (
        wall(lambda: spmm.csr_spmm(rp, ci, val, B, out=C)), wall(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p)),
        wall(lambda: spmm.csr_spmm_no_edge_value(rp, ci, B, out=C, plan=p)),
        timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C), 100), timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=p), 100)))
print("initialised:", _lib._initialised)
