"""Device analysis at scale: forced clustering of RMAT graphs (hubs of 10^5-10^6 entries: the dense-accumulator class),
device order == host order at scale 22, plan bits == plain call.  python scripts/plan_scale_check.py [scale ...]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import gespmm_amd
from gespmm_amd import _lib, graphs, spmm
for scale in [int(a) for a in sys.argv[1:]] or [20, 22, 24]:
    g = graphs.rmat_shard(scale, 16, 0, 1, seed=42, device="cuda")
    rp, ci, M, K = g["rowptr"], g["colind"], g["M"], g["K"]
    N = 64
    val = torch.rand(g["nnz"], device="cuda") - 0.5
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pd = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True, analysis="device", flags=_lib.FLAG_STRICT_ORDER)
    torch.cuda.synchronize(); td = time.perf_counter() - t0
    print("rmat-%d nnz=%d device analysis %.3f s | %s" % (scale, g["nnz"], td, pd.describe()[:200]), flush=True)
    if scale <= 22:
        t0 = time.perf_counter()
        ph = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True, analysis="host", flags=_lib.FLAG_STRICT_ORDER)
        th = time.perf_counter() - t0
        same = np.array_equal(pd.order().numpy(), ph.order().numpy())
        print("   host analysis %.3f s, same order: %s" % (th, same), flush=True)
        del ph
    B = torch.rand(K, N, device="cuda") - 0.5
    ref = spmm.csr_spmm(rp, ci, val, B, cfg={"flags": _lib.FLAG_STRICT_ORDER})
    got = spmm.csr_spmm(rp, ci, val, B, plan=pd)
    print("   bits equal plain strict call:", bool(torch.equal(ref.view(torch.int32), got.view(torch.int32))), flush=True)
    del pd, ref, got, B, val, g
    torch.cuda.empty_cache()
