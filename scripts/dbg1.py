import sys, ctypes, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gespmm_amd
from gespmm_amd import _lib, spmm
rp = torch.tensor([0, 1, 2, 5], dtype=torch.int32, device="cuda")
ci = torch.tensor([0, 1, 0, 1, 2], dtype=torch.int32, device="cuda")
v = torch.ones(5, device="cuda")
for N in (3, 32, 128):
    B = torch.ones(3, N, device="cuda")
    for flags in (0, _lib.FLAG_SPLIT_LONG_ROWS, _lib.FLAG_SLAB_SWEEP, _lib.FLAG_BATCH_STREAM, _lib.FLAG_STRICT_ORDER):
        for variant in (-1, 1, 2, 3, 4):
            try:
                spmm.csr_spmm(rp, ci, v, B, variant=variant, cfg=dict(flags=flags))
                torch.cuda.synchronize()
            except Exception as e:
                print(N, hex(flags), variant, str(e)[:80])
print("done")
