import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gespmm_amd
from gespmm_amd import _lib as F, graphs, spmm
dev = torch.device("cuda:0")
g = graphs.synthetic_graph("reddit-like", device=dev)
val = torch.rand(g["nnz"], device=dev); B = torch.rand(g["K"], 128, device=dev); C = torch.empty(g["M"], 128, device=dev)
for _ in range(3):
    spmm.csr_spmm(g["rowptr"], g["colind"], val, B, out=C, cfg=dict(flags=F.FLAG_SLAB_BLOCKED))
    spmm.csr_spmm(g["rowptr"], g["colind"], val, B, out=C, cfg=dict(flags=F.FLAG_NO_SLAB_BLOCKED))
torch.cuda.synchronize()
