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
    for fl in (-1, 4, 8, 12, 16):
        for te in (32, 48, 64, 96, 128, 192):
            plan=spmm.SpmmPlan(rp,ci,K,N,values=val,reorder=True,task_entries=te,row_floor=fl)
            print(name,"floor",fl,"entries",te,"%.1f us"%timeit(lambda: spmm.csr_spmm(rp,ci,val,B,out=C,plan=plan)), plan.describe().split("|")[0].split("tasks=")[1].split()[0],flush=True)
