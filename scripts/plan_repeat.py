import sys, time, torch
sys.path.insert(0, ".")
import gespmm_amd
from gespmm_amd import graphs, spmm
g = graphs.synthetic_graph("products-sbm", seed=42, device="cuda")
val = torch.rand(g["nnz"], device="cuda") - 0.5
keep = []
for i in range(8):
    N = 128 if i % 2 == 0 else 256
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p = spmm.SpmmPlan(g["rowptr"], g["colind"], g["K"], N, values=val)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("plan %d N=%d: %.1f ms | %s" % (i, N, dt * 1e3, p.describe().split("|")[0][-90:]), flush=True)
    if i in (1, 2, 5): keep.append(p)   # some plans stay alive while the next is made (the bench's situation)
    if i == 3: torch.cuda.empty_cache()
