"""Round 3's plan on the structureless com-Amazon stand-in runs 143 us where round 2's ran 136 us at the same traffic and L2 hit
rate: round 2's library (built from d8d0570 into _build/libgespmm_r02.so) next to the current one, host and
device analysis, 3 and 6 clustering levels; orders and task tables compared."""
import ctypes
import os
import statistics
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import gespmm_amd  # noqa
from gespmm_amd import graphs, _lib

old = ctypes.CDLL(os.path.join(HERE, "_build", "libgespmm_r02.so"))
new = _lib.lib
dev = torch.device("cuda")
vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32


class Opt6(ctypes.Structure):
    _fields_ = [(n, i32) for n in ("reorder", "task_entries", "row_floor", "threads", "flags", "kernel")]


class Opt7(ctypes.Structure):
    _fields_ = [(n, i32) for n in ("reorder", "task_entries", "row_floor", "threads", "flags", "kernel", "analysis")]


for L in (old, new):
    L.gespmm_plan_create.restype = ctypes.c_int
    L.gespmm_plan_create.argtypes = [ctypes.POINTER(vp), vp, vp, vp, i64, i64, i64, i64, ctypes.c_int, vp, vp]
    L.gespmm_plan_spmm_f32.restype = ctypes.c_int
    L.gespmm_plan_spmm_f32.argtypes = [vp, vp, vp, i64, vp]
    L.gespmm_plan_describe.argtypes = [vp, ctypes.c_char_p, i64]
    L.gespmm_plan_get_order.argtypes = [vp, vp]
    L.gespmm_plan_destroy.argtypes = [vp]
new.gespmm_plan_debug_tasks.argtypes = [vp, i32, vp, i64]
new.gespmm_plan_debug_tasks.restype = ctypes.c_int


def describe(L, p):
    buf = ctypes.create_string_buffer(2048)
    L.gespmm_plan_describe(p, buf, 2048)
    return buf.value.decode()


def times(fn, n=200):
    for _ in range(10):
        fn()
    s = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    e = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    torch.cuda.synchronize()
    for i in range(n):
        s[i].record()
        fn()
        e[i].record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) * 1e3 for a, b in zip(s, e))


gname = sys.argv[1] if len(sys.argv) > 1 else "com-amazon-like"
g = graphs.synthetic_graph(gname, seed=42, device=dev)
M, K, nnz = g["M"], g["K"], g["nnz"]
rp, ci = g["rowptr"], g["colind"]
gen = torch.Generator(device=dev)
gen.manual_seed(7)
val = torch.rand(nnz, generator=gen, device=dev) - 0.5
st = vp(torch.cuda.current_stream().cuda_stream)
for N in (128, 32, 512):
    B = torch.rand((K, N), device=dev) - 0.5
    C = torch.empty((M, N), device=dev)
    plans = {}

    def mk(tag, L, optobj, env=None):
        for k, v in (env or {}).items():
            os.environ[k] = v
        p = vp()
        rc = L.gespmm_plan_create(ctypes.byref(p), rp.data_ptr(), ci.data_ptr(), val.data_ptr(), M, K, nnz, N, -1,
                                  ctypes.addressof(optobj) if optobj is not None else None, st)
        for k in (env or {}):
            del os.environ[k]
        assert rc == 0, (tag, rc)
        torch.cuda.synchronize()
        plans[tag] = (L, p)
        print("  [%s] %s" % (tag, describe(L, p)), flush=True)

    print("=== %s N=%d" % (gname, N), flush=True)
    mk("r02 lib (host, 6 levels)", old, None)
    mk("now host", new, Opt7(0, 0, 0, 0, 0, 0, 1))
    mk("now device", new, None)
    mk("now device levels=6", new, None, {"GESPMM_CLUSTER_LEVELS": "6"})
    mk("now host levels=6", new, Opt7(0, 0, 0, 0, 0, 0, 1), {"GESPMM_CLUSTER_LEVELS": "6"})
    mk("now device stream kernel", new, Opt7(0, 0, 0, 0, 0, 1, 0))
    ref = None
    orders = {}
    for tag, (L, p) in plans.items():
        o = np.empty(M, dtype=np.int32)
        L.gespmm_plan_get_order(p, o.ctypes.data)
        orders[tag] = o
    base = orders["r02 lib (host, 6 levels)"]
    for tag, o in orders.items():
        print("  order [%s] == r02's: %s (%.1f %% positions equal)" % (tag, bool((o == base).all()), 100.0 * (o == base).mean()))
    for tag in ("now host", "now device", "now device levels=6"):
        L, p = plans[tag]
        n = new.gespmm_plan_debug_tasks(p, 0, None, 0)
        t = np.empty((n, 4), dtype=np.int32)
        new.gespmm_plan_debug_tasks(p, 0, t.ctypes.data, n)
        print("  tasks [%s]: %d, rows/task mean %.2f max %d, entries/task mean %.1f max %d" %
              (tag, n, t[:, 1].mean(), t[:, 1].max(), (t[:, 3] - t[:, 2]).mean(), (t[:, 3] - t[:, 2]).max()))
    for rep in range(2):
        row = []
        for tag, (L, p) in plans.items():
            f = lambda: L.gespmm_plan_spmm_f32(p, B.data_ptr(), C.data_ptr(), N, st)  # noqa: E731
            assert f() == 0
            torch.cuda.synchronize()
            if ref is None:
                ref = C.clone()
            else:
                assert torch.equal(C.view(torch.int32), ref.view(torch.int32)), tag
            row.append("%s %.1f" % (tag, times(f)))
        print("  us: " + " | ".join(row), flush=True)
    for L, p in plans.values():
        L.gespmm_plan_destroy(p)
