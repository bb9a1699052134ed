import sys; sys.path.insert(0,'.')
import torch, gespmm_amd
from gespmm_amd import graphs, spmm, _lib
dev=torch.device("cuda")
def timeit(fn, iters=200):
    for _ in range(20): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters*1e3
for name in ("com-amazon-sbm","com-amazon-like"):
    g=graphs.synthetic_graph(name,seed=42,device=dev); M,K,nnz=g["M"],g["K"],g["nnz"]; rp,ci=g["rowptr"],g["colind"]
    val=torch.rand(nnz,device=dev)-0.5
    N=128
    B=((torch.randint(0,100,(K,N),device=dev,dtype=torch.int32)-50).float()/100); C=torch.empty((M,N),device=dev)
    plans = {}
    for te in (64, 96, 128):
        plans[("hubs-first",te)] = spmm.SpmmPlan(rp,ci,K,N,values=val,reorder=True,task_entries=te,kernel="stream")
        plans[("hubs-in-place",te)] = spmm.SpmmPlan(rp,ci,K,N,values=val,reorder=True,task_entries=te,kernel="stream",flags=0x10000)
    for rep in range(3):
        print(name,"plain %.1f"%timeit(lambda: spmm.csr_spmm(rp,ci,val,B,out=C)), " ".join("%s/%d %.1f"%(k[0],k[1],timeit(lambda: spmm.csr_spmm(rp,ci,val,B,out=C,plan=p))) for k,p in plans.items()),flush=True)
