"""Fuzz of the device analysis stage: random shapes (tiny, rectangular, empty rows, hubs, duplicates) — device order == host
order, same task tables, SpMM bits == plain call.  python scripts/plan_device_fuzz.py [cases] [seed]"""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gespmm_amd
from gespmm_amd import _lib, spmm
lib = _lib.lib
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
ran = 0
clustered = 0
for case in range(cases):
    M = int(rng.choice([1, 2, 7, 64, 300, 2000, 20000, 60000]))
    K = M if rng.rand() < 0.5 else int(rng.randint(1, 3 * M + 2))
    kind = rng.randint(4)
    if kind == 0: deg = rng.randint(0, 6, M)
    elif kind == 1: deg = (rng.pareto(1.2, M) * 3).astype(np.int64).clip(0, min(K * 2, 30000))
    elif kind == 2: deg = rng.randint(0, 200, M) * (rng.rand(M) < 0.3)
    else:
        deg = rng.randint(1, 12, M); deg[rng.randint(0, M, max(1, M // 500))] = min(K * 3, 9000)
    rp = np.zeros(M + 1, dtype=np.int32); rp[1:] = np.cumsum(deg)
    nnz = int(rp[-1])
    if nnz == 0 or nnz > 6_000_000: continue
    # columns: clustered structure half of the time
    if rng.rand() < 0.5:
        rows = np.repeat(np.arange(M), deg)
        ci = ((rows * K // max(M, 1)) + rng.randint(-20, 21, nnz)).clip(0, K - 1).astype(np.int32)
    else:
        ci = rng.randint(0, K, nnz).astype(np.int32)
    rp_d, ci_d = torch.from_numpy(rp).cuda(), torch.from_numpy(ci).cuda()
    N = int(rng.choice([4, 32, 128]))
    val = torch.rand(nnz, device="cuda") - 0.5
    try:
        pd = spmm.SpmmPlan(rp_d, ci_d, K, N, values=val, reorder=True, analysis="device", flags=_lib.FLAG_STRICT_ORDER)
        ph = spmm.SpmmPlan(rp_d, ci_d, K, N, values=val, reorder=True, analysis="host", flags=_lib.FLAG_STRICT_ORDER)
    except Exception as ex:
        print("case %d M=%d K=%d nnz=%d kind=%d: plan failed: %s" % (case, M, K, nnz, kind, ex)); bad += 1; continue
    ok = True
    ran += 1
    clustered += int(pd.clustered)
    if pd.clustered != ph.clustered: ok = False
    if pd.clustered and not np.array_equal(pd.order().numpy(), ph.order().numpy()): ok = False
    for which in (0, 1):
        n1 = lib.gespmm_plan_debug_tasks(pd._handle, which, None, 0); n2 = lib.gespmm_plan_debug_tasks(ph._handle, which, None, 0)
        if n1 != n2: ok = False; break
        a = np.empty((max(n1, 1), 4), dtype=np.int32); b = np.empty((max(n1, 1), 4), dtype=np.int32)
        lib.gespmm_plan_debug_tasks(pd._handle, which, a.ctypes.data, n1); lib.gespmm_plan_debug_tasks(ph._handle, which, b.ctypes.data, n1)
        if not np.array_equal(a[:n1], b[:n1]): ok = False
    B = torch.rand(K, N, device="cuda") - 0.5
    ref = spmm.csr_spmm(rp_d, ci_d, val, B, cfg={"flags": _lib.FLAG_STRICT_ORDER})
    got = spmm.csr_spmm(rp_d, ci_d, val, B, plan=pd)
    if not torch.equal(ref.view(torch.int32), got.view(torch.int32)): ok = False
    if not ok:
        bad += 1
        print("case %d M=%d K=%d nnz=%d kind=%d N=%d: MISMATCH  %s | %s" % (case, M, K, nnz, kind, N, pd.describe()[:80], ph.describe()[:80]))
print("%d cases drawn, %d run (%d clustered), %d bad" % (cases, ran, clustered, bad))
