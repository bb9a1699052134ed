import sys; sys.path.insert(0,'.')
import torch, gespmm_amd
from gespmm_amd import graphs, spmm, sddmm
dev=torch.device("cuda")
def timeit(fn, iters=100):
    for _ in range(10): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters*1e3
for name in ("com-amazon-sbm","com-amazon-like"):
    g=graphs.synthetic_graph(name,seed=42,device=dev); M,K=g["M"],g["K"]; rp,ci=g["rowptr"],g["colind"]
    rows=torch.repeat_interleave(torch.arange(M,device=dev,dtype=torch.int32),(rp[1:]-rp[:-1]).long())
    for N in (128, 41):
        D1=torch.rand(M,N,device=dev)-0.5; D2=torch.rand(K,N,device=dev)-0.5
        plan=spmm.SpmmPlan(rp,ci,K,N,reorder=True)
        print(name,"N",N,"coo %.1f us  csr %.1f us  planned %.1f us"%(timeit(lambda: sddmm.coo_sddmm(rows,ci,D1,D2)),timeit(lambda: sddmm.csr_sddmm(rp,ci,D1,D2)),timeit(lambda: sddmm.csr_sddmm(rp,ci,D1,D2,plan=plan))),flush=True)
