import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, gespmm_amd
from gespmm_amd import _lib as F, graphs, spmm
dev=torch.device("cuda:0")
def time_fn(fn, iters=300, warm=30):
    for _ in range(warm): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters*1e3
for name in ("pubmed-like","cit-hepth-like","com-amazon-like"):
    g=graphs.synthetic_graph(name,device=dev); rp,ci,M,K=g["rowptr"],g["colind"],g["M"],g["K"]
    val=torch.rand(ci.numel(),device=dev)
    for N in (32,128):
        B=torch.rand(K,N,device=dev); C=torch.empty(M,N,device=dev)
        a=time_fn(lambda: spmm.csr_spmm(rp,ci,val,B,out=C))
        b=time_fn(lambda: spmm.csr_spmm(rp,ci,val,B,out=C,cfg=dict(flags=F.FLAG_SPLIT_LONG_ROWS)))
        print("%s N=%d: auto %.1f us, with the long-row pass forced %.1f us (+%.1f)"%(name,N,a,b,b-a))
