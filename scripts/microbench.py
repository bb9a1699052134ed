#!/usr/bin/env python3
"""Runs scripts/microbench.hip: streaming-read and random-row-gather ceilings of the
memory hierarchy (L2 / Infinity Cache / HBM footprints)."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = "/tmp/libmb.so"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                os.path.join(ROOT, "scripts", "microbench.hip"), "-o", so], check=True)
lib = ctypes.CDLL(so)
P = ctypes.c_void_p
lib.mb_stream_read.argtypes = [P, ctypes.c_size_t, P, ctypes.c_int, P]
lib.mb_row_gather.argtypes = [P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, ctypes.c_int, P]
lib.mb_row_gather128.argtypes = [P, P, ctypes.c_int, ctypes.c_int, P, ctypes.c_int, P]


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


dev = torch.device("cuda:0")
st = P(torch.cuda.current_stream().cuda_stream)
out = torch.zeros(1024, device=dev)
print("== streaming read (dwordx4, grid-stride), best over block counts")
for mb in (2, 16, 64, 128, 171, 200, 343, 1024, 4096):
    x = torch.rand(mb * 250000, device=dev)
    best = None
    for blocks in (1024, 2048, 4096, 8192):
        us = timeit(lambda: lib.mb_stream_read(P(x.data_ptr()), x.numel() * 4, P(out.data_ptr()), blocks, st))
        best = us if best is None else min(best, us)
    print("  %5d MB: %8.1f us  %.2f TB/s" % (mb, best, mb / best))
    del x
print("== random 512-B row gather, nidx = 1 851 744 (948 MB gathered), store every 5.5 rows ~ SpMM C traffic")
nidx = 1851744
for K in (2048, 8192, 32768, 131072, 334863, 1 << 20, 1 << 22):
    B = torch.rand((K, 128), device=dev)
    idx = torch.randint(0, K, (nidx,), device=dev, dtype=torch.int32)
    C = torch.empty((nidx // 5 + 8, 128), device=dev)
    for per_store, tag in ((0, "no store"), (8, "store/8 rows")):
        best = None
        for blocks in (2048, 4096, 8192, 16384):
            us = timeit(lambda: lib.mb_row_gather(P(B.data_ptr()), P(idx.data_ptr()), nidx, 128, per_store,
                                                  P(C.data_ptr()), blocks, st))
            best = us if best is None else min(best, us)
        print("  K=%8d (B %7.1f MB) %-13s: %8.1f us  gather %.2f TB/s" %
              (K, K * 512 / 1e6, tag, best, nidx * 512 / best / 1e6))
    del B, idx, C

print("== random 128-B row gather (N = 32), nidx = 1 851 744 (237 MB gathered)")
for K in (8192, 334863, 1 << 22):
    B = torch.rand((K, 32), device=dev)
    idx = torch.randint(0, K, (nidx,), device=dev, dtype=torch.int32)
    C = torch.empty((nidx // 5 + 8, 32), device=dev)
    for per_store, tag in ((0, "no store"), (8, "store/8 rows")):
        best = None
        for blocks in (1024, 2048, 4096, 8192):
            us = timeit(lambda: lib.mb_row_gather128(P(B.data_ptr()), P(idx.data_ptr()), nidx, per_store,
                                                     P(C.data_ptr()), blocks, st))
            best = us if best is None else min(best, us)
        print("  K=%8d (B %7.1f MB) %-13s: %8.1f us  gather %.2f TB/s" %
              (K, K * 128 / 1e6, tag, best, nidx * 128 / best / 1e6))
    del B, idx, C
