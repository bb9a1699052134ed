"""EXPERIMENT: cooperative slab sweep (csrc/spmm_slabsweep.hip, gespmm_debug_slabsweep_f32) vs the shipped cache-blocked path
on the reddit-shaped graph.  python profiles/r03/experiments/slab_sweep_time.py [scale]"""
import ctypes, statistics, sys
import torch
sys.path.insert(0, ".")
import gespmm_amd
from gespmm_amd import _lib, graphs, spmm
lib = _lib.lib
fn = lib.gespmm_debug_slabsweep_f32
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int64] * 4 + [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
g = graphs.synthetic_graph("reddit-like", seed=42, device="cuda", scale=scale)
rp, ci, M, K = g["rowptr"], g["colind"], g["M"], g["K"]
val = torch.rand(g["nnz"], device="cuda") - 0.5
for N in (128, 64):
    B = torch.rand(K, N, device="cuda") - 0.5
    plan = spmm.SpmmPlan(rp, ci, K, N, values=val)
    ref = spmm.csr_spmm(rp, ci, val, B, plan=plan)
    def med(f, n=7):
        for _ in range(2): f()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in ev:
            a.record(); f(); b.record()
        torch.cuda.synchronize()
        return statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev)
    t_ship = med(lambda: spmm.csr_spmm(rp, ci, val, B, plan=plan, out=ref))
    print("N=%d shipped cache-blocked path (plan): %.1f us | %s" % (N, t_ship, plan.describe()[-90:]), flush=True)
    for slab_mb in (1.5, 2, 3, 4, 6):
        slab_rows = int(slab_mb * (1 << 20) / (N * 4))
        nslab = (K + slab_rows - 1) // slab_rows
        ws = torch.empty((nslab + 1) * M + 16, dtype=torch.int32, device="cuda")
        C = torch.empty(M, N, device="cuda")
        rc = fn(rp.data_ptr(), ci.data_ptr(), val.data_ptr(), B.data_ptr(), C.data_ptr(), M, K, N, slab_rows, ws.data_ptr(), 1, None)
        torch.cuda.synchronize()
        if rc < 0:
            print("  slab %.1f MB: rc %d" % (slab_mb, rc), flush=True); continue
        same = bool(torch.equal(C.view(torch.int32), ref.view(torch.int32)))
        t = med(lambda: fn(rp.data_ptr(), ci.data_ptr(), val.data_ptr(), B.data_ptr(), C.data_ptr(), M, K, N, slab_rows, ws.data_ptr(), 0, None))
        print("  cooperative sweep, slab %.1f MB (%d slabs, %d workgroups): %.1f us  bits equal shipped path: %s" % (slab_mb, nslab, rc, t, same), flush=True)
