import sys, time; sys.path.insert(0,'.')
import torch, gespmm_amd
from gespmm_amd import graphs, spmm
dev=torch.device("cuda")
def timeit(fn, iters=10):
    for _ in range(2): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters
g=graphs.synthetic_graph("products-sbm",seed=42,device=dev); M,K,nnz=g["M"],g["K"],g["nnz"]; rp,ci=g["rowptr"],g["colind"]
val=torch.rand(nnz,device=dev)-0.5
for N in (128, 64, 256):
    B=torch.rand(K,N,device=dev)-0.5; C=torch.empty((M,N),device=dev)
    ref=spmm.csr_spmm(rp,ci,val,B).clone()
    for kern in ("stream","seg-stream","lds-rows"):
        for te in ((0,) if kern=="lds-rows" else (0,64,128,256)):
            plan=spmm.SpmmPlan(rp,ci,K,N,values=val,reorder=True,kernel=kern,task_entries=te)
            ms=timeit(lambda: spmm.csr_spmm(rp,ci,val,B,out=C,plan=plan))
            print("N",N,kern,"entries",te,"%.3f ms"%ms,"bits",torch.equal(C.view(torch.int32),ref.view(torch.int32)),plan.describe().split("|")[1][:90],flush=True)
            del plan
