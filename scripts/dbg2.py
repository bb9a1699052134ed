import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, gespmm_amd
from gespmm_amd import _lib, spmm
from test_gpu_spmm import _skewed_csr
G, long_rows = _skewed_csr()
rp = torch.from_numpy(G["rowptr"]).cuda(); ci = torch.from_numpy(G["colind"]).cuda()
v = torch.ones(G["nnz"], device="cuda")
print("M", G["M"], "K", G["K"], "nnz", G["nnz"])
for N in (3, 32, 128, 200, 512):
    B = torch.ones(G["K"], N, device="cuda")
    for flags in (0, _lib.FLAG_SPLIT_LONG_ROWS, _lib.FLAG_STRICT_ORDER, _lib.FLAG_BATCH_STREAM, _lib.FLAG_NO_SLAB_BLOCKED, _lib.FLAG_NO_SLAB_BLOCKED | _lib.FLAG_SPLIT_LONG_ROWS):
        for variant in (-1, 1, 2, 3, 4):
            try:
                spmm.csr_spmm(rp, ci, v, B, variant=variant, cfg=dict(flags=flags)); torch.cuda.synchronize()
                pass
            except Exception as e:
                print(N, hex(flags), variant, str(e)[:70])
