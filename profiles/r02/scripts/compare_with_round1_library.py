"""Plain gespmm_csr_spmm_f32 of the round-1 library vs the current one on the same graphs (regression check)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import gespmm_amd  # noqa
from gespmm_amd import graphs, _lib

old = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgespmm_r01.so"))
new = _lib.lib
dev = torch.device("cuda")
sig = [ctypes.c_void_p] * 5 + [ctypes.c_int64] * 4 + [ctypes.c_int, ctypes.c_void_p]
for L in (old, new):
    L.gespmm_csr_spmm_f32.restype = ctypes.c_int
    L.gespmm_csr_spmm_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]


def timeit(fn, iters):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


cases = [("com-amazon-like", 128, 200), ("com-amazon-like", 32, 200), ("com-amazon-like", 512, 100), ("products-like", 128, 5), ("reddit-like", 128, 5), ("cit-hepth-like", 32, 200)]
for name, N, iters in cases:
    g = graphs.synthetic_graph(name, seed=42, device=dev)
    M, K, nnz = g["M"], g["K"], g["nnz"]
    rp, ci = g["rowptr"], g["colind"]
    val = torch.rand(nnz, device=dev) - 0.5
    B = torch.rand((K, N), device=dev); C = torch.empty((M, N), device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    res = {}
    for tag, L in (("r01", old), ("now", new), ("r01 again", old), ("now again", new)):
        f = lambda: L.gespmm_csr_spmm_f32(rp.data_ptr(), ci.data_ptr(), val.data_ptr(), B.data_ptr(), C.data_ptr(), M, K, N, nnz, -1, st)
        assert f() == 0
        res[tag] = timeit(f, iters)
    print("%-16s N=%-4d " % (name, N) + " | ".join("%s %.1f us" % kv for kv in res.items()), flush=True)
for scale, N in ((22, 256), (24, 256), (22, 128)):
    g = graphs.rmat_shard(scale, 16, 0, 1, seed=42, device=dev)
    M, K, nnz = g["M"], g["K"], g["nnz"]
    rp, ci = g["rowptr"], g["colind"]
    val = torch.rand(nnz, device=dev) - 0.5
    B = torch.rand((K, N), device=dev); C = torch.empty((M, N), device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    res = {}
    for tag, L in (("r01", old), ("now", new), ("r01 again", old), ("now again", new)):
        f = lambda: L.gespmm_csr_spmm_f32(rp.data_ptr(), ci.data_ptr(), val.data_ptr(), B.data_ptr(), C.data_ptr(), M, K, N, nnz, -1, st)
        assert f() == 0
        res[tag] = timeit(f, 5)
    print("rmat-%d N=%-4d " % (scale, N) + " | ".join("%s %.1f us" % kv for kv in res.items()), flush=True)
