#!/usr/bin/env python3
"""Round 6, RMAT shard x N = 256: would running the long-row pass CONCURRENTLY with the main kernel help (round-5 review, item 1)? The
cheapest decisive test needs no library change: two independent products of the same shard on two streams. If the memory system had room
while one product runs, the pair would finish in well under twice the time of one; if the product is bound by the rate at which the
memory system serves its misses, the pair takes twice as long and overlapping two of its kernels cannot gain either.
Also: the product's two big kernels timed separately (rocprofv3 --kernel-trace --stats of this script gives the same split)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from gespmm_amd import graphs, spmm  # noqa: E402

dev = torch.device("cuda")
N = 256
for scale in [int(s) for s in sys.argv[1:]] or [22, 24]:
    g = graphs.rmat_shard(scale, device=dev)
    M, K, nnz, rp, ci = g["M"], g["K"], g["nnz"], g["rowptr"], g["colind"]
    val = torch.rand(nnz, device=dev) - 0.5
    B = torch.rand(K, N, device=dev) - 0.5
    C1 = torch.empty((M, N), device=dev)
    C2 = torch.empty((M, N), device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    iters = 4

    def one(stream, C):
        with torch.cuda.stream(stream):
            spmm.csr_spmm(rp, ci, val, B, out=C)

    for _ in range(2):
        one(s1, C1)
        one(s2, C2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s1)
    for _ in range(iters):
        one(s1, C1)
    e1.record(s1)
    torch.cuda.synchronize()
    t_one = e0.elapsed_time(e1) / iters
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        one(s1, C1)
        one(s2, C2)
    torch.cuda.synchronize()
    t_pair = (time.perf_counter() - t0) * 1e3 / iters
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        one(s1, C1)
        one(s1, C2)
    torch.cuda.synchronize()
    t_seq = (time.perf_counter() - t0) * 1e3 / iters
    print("rmat-%d x N=%d: one product %.2f ms | two products on ONE stream %.2f ms | two products on TWO streams %.2f ms (x%.2f of one)"
          % (scale, N, t_one, t_seq, t_pair, t_pair / t_one), flush=True)
    assert torch.equal(C1, C2)
    del g, rp, ci, val, B, C1, C2
    torch.cuda.empty_cache()
