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
    for variant in (3, 4, 2):
        for te in (64, 128, 192):
            for fl,lab in ((0,""),(_lib.FLAG_SHALLOW_UNROLL,"shallow")):
                plan=spmm.SpmmPlan(rp,ci,K,N,variant=variant,values=val,reorder=True,task_entries=te,kernel="stream",flags=fl)
                print(name,"variant",variant,"entries",te,lab,"%.1f us"%timeit(lambda: spmm.csr_spmm(rp,ci,val,B,variant=variant,out=C,plan=plan)), plan.describe().split("|")[1][:70],flush=True)
