#!/usr/bin/env python3
"""Plans on the bench graphs: plain call vs storage-order plan vs clustered plan, task-size sweep.

    python scripts/plan_bench.py [--graphs com-amazon-sbm,com-amazon-like] [--ncols 128] [--entries 16,24,32,48,64,96]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", default="com-amazon-sbm,com-amazon-like")
    ap.add_argument("--ncols", default="128")
    ap.add_argument("--entries", default="0,16,24,32,48,64,96,128")
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--kernel", default="auto", choices=("auto", "stream", "seg-stream", "staged"))
    ap.add_argument("--only-plan", action="store_true", help="one clustered plan per graph at the default task size (for rocprofv3)")
    ap.add_argument("--task-entries", type=int, default=0, help="with --only-plan: task size (0 = default)")
    args = ap.parse_args()
    import torch

    import gespmm_amd  # noqa: F401
    from gespmm_amd import graphs, spmm

    dev = torch.device("cuda")

    def timeit(fn):
        for _ in range(20):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.iters * 1e3

    for name in args.graphs.split(","):
        g = graphs.synthetic_graph(name, seed=42, device=dev)
        M, K, nnz = g["M"], g["K"], g["nnz"]
        rp, ci = g["rowptr"], g["colind"]
        val = torch.rand(nnz, device=dev) - 0.5
        for N in [int(x) for x in args.ncols.split(",")]:
            B = ((torch.randint(0, 100, (K, N), device=dev, dtype=torch.int32) - 50).float() / 100)
            C = torch.empty((M, N), device=dev)
            abytes = 4 * (M + 1) + 8 * nnz + 4 * K * N + 4 * M * N
            if args.only_plan:
                plan = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True, kernel=args.kernel, task_entries=args.task_entries)
                us = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan))
                print("%s N=%d clustered plan %.1f us  frac %.3f | %s" % (name, N, us, abytes / us / 8e6, plan.describe()), flush=True)
                continue
            us = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C))
            ref = C.clone()
            print("%s N=%d plain AUTO            %8.1f us  frac %.3f" % (name, N, us, abytes / us / 8e6), flush=True)
            plan = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=False)
            us = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan))
            print("%s N=%d storage-order plan    %8.1f us  frac %.3f" % (name, N, us, abytes / us / 8e6), flush=True)
            for kernel in ("stream", "seg-stream"):
                for te in [int(x) for x in args.entries.split(",")]:
                    t0 = time.time()
                    plan = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True, task_entries=te, kernel=kernel)
                    dt = time.time() - t0
                    C.zero_()
                    us = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan))
                    ok = torch.equal(C.view(torch.int32), ref.view(torch.int32))
                    print("%s N=%d clustered %-8s entries=%-4d %8.1f us  frac %.3f  bits_equal=%s  create %.2fs" %
                          (name, N, kernel, te, us, abytes / us / 8e6, ok, dt), flush=True)
                print("   ", plan.describe(), flush=True)
            auto = spmm.SpmmPlan(rp, ci, K, N, values=val)
            us = timeit(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=auto))
            print("%s N=%d AUTO plan             %8.1f us  | %s" % (name, N, us, auto.describe()), flush=True)


if __name__ == "__main__":
    main()
