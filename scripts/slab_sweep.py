#!/usr/bin/env python3
"""Cache-blocked (slab) path on the reddit-shaped graph: lanes per row (column tile) x slab height x rows per task.
    python scripts/slab_sweep.py [--ncols 128] [--groups 32,16] [--slab-rows 0,6144,...] [--rpw 0,2,4]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ncols", type=int, default=128)
    ap.add_argument("--groups", default="0,32,16")
    ap.add_argument("--slab-rows", default="0,6144,8192,12288,16384,24576")
    ap.add_argument("--rpw", default="0")
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    import torch

    import gespmm_amd  # noqa: F401
    from gespmm_amd import _lib, graphs, spmm

    dev = torch.device("cuda")
    g = graphs.synthetic_graph("reddit-like", seed=42, device=dev)
    M, K, nnz = g["M"], g["K"], g["nnz"]
    rp, ci = g["rowptr"], g["colind"]
    val = torch.rand(nnz, device=dev) - 0.5
    N = args.ncols
    B = torch.rand(K, N, device=dev) - 0.5
    C = torch.empty((M, N), device=dev)
    ref = spmm.csr_spmm(rp, ci, val, B, cfg={"flags": _lib.FLAG_NO_SLAB_BLOCKED | _lib.FLAG_STRICT_ORDER}).clone()
    for grp in [int(x) for x in args.groups.split(",")]:
        for sr in [int(x) for x in args.slab_rows.split(",")]:
            for rpw in [int(x) for x in args.rpw.split(",")]:
                cfg = {"group": grp, "slab_rows": sr, "rows_per_wave": rpw, "flags": _lib.FLAG_SLAB_BLOCKED}
                try:
                    for _ in range(2):
                        spmm.csr_spmm(rp, ci, val, B, cfg=cfg, out=C)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(args.iters):
                        spmm.csr_spmm(rp, ci, val, B, cfg=cfg, out=C)
                    e1.record()
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / args.iters
                    ok = torch.equal(C.view(torch.int32), ref.view(torch.int32))
                    print("N=%d group=%-2d slab_rows=%-6d rpw=%d  %7.3f ms  %6.2f TFLOP/s  bits_equal=%s" %
                          (N, grp, sr, rpw, ms, 2.0 * nnz * N / ms / 1e9, ok), flush=True)
                except Exception as ex:  # noqa: BLE001
                    print("N=%d group=%d slab_rows=%d rpw=%d failed: %s" % (N, grp, sr, rpw, ex), flush=True)


if __name__ == "__main__":
    main()
