#!/usr/bin/env python3
"""Round-over-round regression check that does not depend on the box: the PREVIOUS round's libgespmm.so (built from its commit into
profiles/rNN/experiments/_build/, git-ignored, travels with the snapshot) is loaded next to the current one and both run AUTO plans
and plain calls on the same operands in ONE process, interleaved twice (boxes differ by +-4 % in absolute time, two libraries in one
process by +-0.5 %). A (graph, N) whose plan or plain call is more than 2 % slower than the old library's is flagged; exit 1.

    python scripts/plan_regression_ab.py profiles/r04/experiments/_build/libgespmm_r03.so

To make the old library (no GPU needed):
    git archive <rev> gespmm_amd/csrc include | tar -x -C /tmp/old && make -C /tmp/old/gespmm_amd/csrc -j8 ../lib/libgespmm.so
"""
import ctypes
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import gespmm_amd  # noqa: F401,E402
from gespmm_amd import _lib, graphs  # noqa: E402

old = ctypes.CDLL(os.path.abspath(sys.argv[1]))
new = _lib.lib
vp, i64 = ctypes.c_void_p, ctypes.c_int64
for L in (old,):  # (the current library's prototypes are set by gespmm_amd._lib)
    L.gespmm_plan_create.restype = ctypes.c_int
    L.gespmm_plan_create.argtypes = [ctypes.POINTER(vp), vp, vp, vp, i64, i64, i64, i64, ctypes.c_int, vp, vp]
    L.gespmm_plan_spmm_f32.restype = ctypes.c_int
    L.gespmm_plan_spmm_f32.argtypes = [vp, vp, vp, i64, vp]
    L.gespmm_plan_describe.argtypes = [vp, ctypes.c_char_p, i64]
    L.gespmm_plan_destroy.argtypes = [vp]
    L.gespmm_csr_spmm_f32.restype = ctypes.c_int
    L.gespmm_csr_spmm_f32.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i64, ctypes.c_int, vp]
dev = torch.device("cuda")


def med(fn, n):
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev)


def describe(L, p):
    buf = ctypes.create_string_buffer(2048)
    L.gespmm_plan_describe(p, buf, 2048)
    return buf.value.decode()


CASES = [("com-amazon-sbm", 1.0, (32, 128, 256, 512)), ("com-amazon-like", 1.0, (32, 128, 512)), ("products-sbm", 0.25, (32, 64, 128, 256)),
         ("products-sbm", 1.0, (128, 256, 512)), ("products-like", 0.25, (128,)), ("reddit-sbm", 1.0, (128,)), ("reddit-like", 1.0, (128,)),
         ("pubmed-like", 1.0, (128,)), ("cit-hepth-like", 1.0, (32,))]
flagged = []
for name, scale, widths in CASES:
    g = graphs.synthetic_graph(name, seed=42, device=dev, scale=scale)
    rp, ci, K, M, nnz = g["rowptr"], g["colind"], g["K"], g["M"], g["nnz"]
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    val = torch.rand(nnz, generator=gen, device=dev) - 0.5
    st = vp(torch.cuda.current_stream().cuda_stream)
    for N in widths:
        if 8.0 * (M + K) * N > 60e9:
            continue
        B = torch.rand((K, N), device=dev) - 0.5
        C = torch.empty((M, N), device=dev)
        plans = {}
        for tag, L in (("old", old), ("new", new)):
            p = vp()
            rc = L.gespmm_plan_create(ctypes.byref(p), rp.data_ptr(), ci.data_ptr(), val.data_ptr(), M, K, nnz, N, -1, None, st)
            assert rc == 0, (tag, rc)
            plans[tag] = (L, p)
        torch.cuda.synchronize()
        n = 100 if nnz < 5e6 else (20 if nnz < 5e7 else 8)
        res = {"old plan": [], "new plan": [], "old plain": [], "new plain": []}
        ref = None
        for _ in range(2):
            for tag, (L, p) in plans.items():
                res[tag + " plan"].append(med(lambda: L.gespmm_plan_spmm_f32(p, B.data_ptr(), C.data_ptr(), N, st), n))
                if ref is None:
                    ref = C.clone()
                same = torch.equal(C.view(torch.int32), ref.view(torch.int32))
                res[tag + " plain"].append(med(lambda: L.gespmm_csr_spmm_f32(rp.data_ptr(), ci.data_ptr(), val.data_ptr(), B.data_ptr(),
                                                                             C.data_ptr(), M, K, N, nnz, -1, st), n))
        t = {k: min(v) for k, v in res.items()}
        flag = ""
        for what in ("plan", "plain"):
            if nnz >= (1 << 20) and t["new " + what] > 1.02 * t["old " + what]:
                flag += "  <-- REGRESSION (%s): %.1f vs %.1f us" % (what, t["new " + what], t["old " + what])
                flagged.append((name, scale, N, what))
        print("%-16s x%.2f N=%-3d plan old %9.1f new %9.1f (x%.3f)  plain old %9.1f new %9.1f (x%.3f)  bits %s | new: %s%s" %
              (name, scale, N, t["old plan"], t["new plan"], t["old plan"] / t["new plan"], t["old plain"], t["new plain"],
               t["old plain"] / t["new plain"], "same" if same else "differ (long-row pass geometry)", describe(new, plans["new"][1]).split("|")[-1].strip()[:60],
               flag), flush=True)
        for L, p in plans.values():
            L.gespmm_plan_destroy(p)
        del B, C
    del g
    torch.cuda.empty_cache()
print("# %d regression(s) against %s %s" % (len(flagged), sys.argv[1], flagged or ""))
sys.exit(1 if flagged else 0)
