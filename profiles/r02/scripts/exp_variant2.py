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
for name in ("com-amazon-sbm",):
    g=graphs.synthetic_graph(name,seed=42,device=dev); M,K,nnz=g["M"],g["K"],g["nnz"]; rp,ci=g["rowptr"],g["colind"]
    val=torch.rand(nnz,device=dev)-0.5
    for N in (128, 64, 256, 512):
        B=((torch.randint(0,100,(K,N),device=dev,dtype=torch.int32)-50).float()/100); C=torch.empty((M,N),device=dev)
        for variant in (3, 4):
            for fl,lab in ((0,"U8"),(_lib.FLAG_SHALLOW_UNROLL,"U4")):
                if variant == 4 and fl: continue
                row = []
                for te in (32, 48, 64, 80, 96, 128):
                    for floor in (8,):
                        plan=spmm.SpmmPlan(rp,ci,K,N,variant=variant,values=val,reorder=True,task_entries=te,kernel="stream",flags=fl,row_floor=floor)
                        row.append("%d:%.1f"%(te,timeit(lambda: spmm.csr_spmm(rp,ci,val,B,variant=variant,out=C,plan=plan))))
                print(name,"N",N,"variant",variant,lab," ".join(row),flush=True)
