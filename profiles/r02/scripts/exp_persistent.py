import sys, os; sys.path.insert(0,'.')
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
P=_lib.FLAG_PERSISTENT_TASKS; SH=_lib.FLAG_SHALLOW_UNROLL; NOSH=0x20000
for name in (sys.argv[1:] or ["com-amazon-sbm","com-amazon-like"]):
    g=graphs.synthetic_graph(name,seed=42,device=dev); M,K,nnz=g["M"],g["K"],g["nnz"]; rp,ci=g["rowptr"],g["colind"]
    val=torch.rand(nnz,device=dev)-0.5
    N=128
    B=((torch.randint(0,100,(K,N),device=dev,dtype=torch.int32)-50).float()/100); C=torch.empty((M,N),device=dev)
    ref=spmm.csr_spmm(rp,ci,val,B).clone()
    for lab,fl in (("one-task U8",NOSH),("one-task U4",SH),("persistent U8",P|NOSH),("persistent U4",P|SH)):
        row=[]
        for te in (12,16,24,32,48,64,96):
            plan=spmm.SpmmPlan(rp,ci,K,N,values=val,reorder=True,task_entries=te,kernel="stream",flags=fl)
            C.zero_()
            us=timeit(lambda: spmm.csr_spmm(rp,ci,val,B,out=C,plan=plan))
            ok=torch.equal(C.view(torch.int32),ref.view(torch.int32))
            row.append("%d:%.1f%s"%(te,us,"" if ok else "(BITS!)"))
        print(name,os.environ.get("GESPMM_PERSIST_WGS","-"),lab," ".join(row),flush=True)
