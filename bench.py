#!/usr/bin/env python3
"""bench.py — SpMM GFLOP/s (= 2*nnz*N/t) and achieved HBM GB/s against the roofline.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE launch of the hot path (C = A @ B, valued CSR x dense fp32) over one
resident batch of synthetic input — exactly what the reference's driver times 200x per
width (spmm_test.cu:754-762). Workload at N=1: BASELINE.json configs[1], the
com-Amazon-shaped graph (M = K = 334 863, nnz = 1 851 744) at feature width 128, as a
seeded synthetic stand-in (no network: SURVEY.md §8 d4). Inputs are resident in HBM
before the timed region.

Multi-GPU (weak scaling, one process per GPU): the path is row-partitioned — every rank
owns an equally sized row shard of a tall A ((world*M) x K) with its own seed, and the
dense B (K x N) is replicated by ONE RCCL broadcast before the timed region (reported
as exchange_ms; it is the path's only exchange step, there is no reduction). The timed
region is K SpMM launches per rank on the resident operands; value = total FLOP of all
ranks / max-over-ranks time.

`--graph rmat --rmat-scale S --ncols 256` is the north_star's strong-scaling case: ONE
RMAT graph (Graph500 parameters, S = 26 for the billion-edge run), nnz-balanced
contiguous row shards generated rank-locally from a shared counter-based stream, full
B on every rank; reported as "scaling": "strong".

The JSON line also carries
  roofline      algorithmic bytes / average kernel duration measured
                with HIP events on the launch stream, against 8 TB/s HBM;
  cpu_baseline  the oracle's restatement of the reference's CPU loop
                (spmm_test.cu:595-605) timed on this box's host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
FP32_VALU_PEAK_TFLOPS = 157.3


def algorithmic_bytes(M, K, N, nnz, valued=True):
    """rowptr + colind (+ val) + B read once + C written once."""
    return 4 * (M + 1) + 4 * nnz + (4 * nnz if valued else 0) + 4 * K * N + 4 * M * N


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--ncols", type=int, default=128, help="feature width N of the headline measurement")
    ap.add_argument("--graph", default="com-amazon-like",
                    help="named stand-in (weak scaling: one per rank) or 'rmat' (one fixed graph, strong scaling)")
    ap.add_argument("--rmat-scale", type=int, default=22, help="log2(vertices) of the RMAT graph (north_star: 26)")
    ap.add_argument("--edge-factor", type=int, default=16)
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--locality", type=float, default=0.0, help="fraction of id-local edges in the stand-in")
    ap.add_argument("--no-extra", action="store_true", help="skip the N=32/512 and unweighted side measurements")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import gespmm_amd
    from gespmm_amd import graphs, spmm

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: there is no CPU path to measure")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ  # torchrun with 1 process still exercises RCCL
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)

    # ------------------------------------------------------------------ workload
    strong = args.graph == "rmat"
    if strong:
        # ONE fixed graph, nnz-balanced contiguous row shards: the north_star's
        # "row-partitioned billion-edge synthetic graph" (scale 26) at a selectable scale.
        g = graphs.rmat_shard(args.rmat_scale, args.edge_factor, rank, world, seed=42, device=dev)
    else:
        g = graphs.synthetic_graph(args.graph, seed=42 + rank, device=dev, locality=args.locality)
    M, K, nnz = g["M"], g["K"], g["nnz"]
    rowptr, colind = g["rowptr"], g["colind"]
    gen = torch.Generator(device=dev)
    gen.manual_seed(7 + rank)
    val = torch.rand(nnz, generator=gen, device=dev) - 0.5
    gen_val = val

    widths = [args.ncols] if (args.no_extra or world > 1) else sorted({32, 128, 512, args.ncols})
    maxN = max(widths)

    def make_B(N):
        gB = torch.Generator(device=dev)
        gB.manual_seed(1000 + N)
        # reference value set: float(r % 100 - 50) / 100 (spmm_test.cu:586-594)
        return (torch.randint(0, 100, (K, N), generator=gB, device=dev, dtype=torch.int32) - 50).float() / 100

    exchange_ms = None

    def get_B(N):
        nonlocal exchange_ms
        if not use_dist:
            return make_B(N)
        from gespmm_amd import dist as gdist

        B0 = make_B(N) if rank == 0 else None
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        B = gdist.broadcast_dense(B0, K, N, src=0, device=dev)  # RCCL over xGMI: the path's only exchange
        torch.cuda.synchronize()
        dist.barrier()
        exchange_ms = (time.perf_counter() - t0) * 1e3
        return B

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def measure(N, valued, steps, warmup, variant):
        B = get_B(N)
        C = torch.empty((M, N), dtype=torch.float32, device=dev)
        v = val if valued else None

        def step():
            if valued:
                spmm.csr_spmm(rowptr, colind, v, B, variant=variant, out=C)
            else:
                spmm.csr_spmm_no_edge_value(rowptr, colind, B, variant=variant, out=C)

        for _ in range(warmup):
            step()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        sync_all()
        t0 = time.perf_counter()
        e0.record()  # on the current stream == the stream the C ABI launches on
        for _ in range(steps):
            step()
        e1.record()
        sync_all()
        wall = time.perf_counter() - t0
        kern_ms = e0.elapsed_time(e1) / steps
        if use_dist:
            t = torch.tensor([wall], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall = float(t.item())
        return {"wall_s": wall, "kernel_ms": kern_ms, "B": B, "C": C}

    def verify(B, C, valued, graph=None):
        """Sampled rows against the CPU oracle (checker only, outside the timed region)."""
        rowptr, colind, val, M = graph if graph is not None else (g["rowptr"], g["colind"], gen_val, g["M"])
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import numpy as np

        import oracle_py

        rng = np.random.RandomState(0)
        rows = np.sort(rng.choice(M, min(512, M), replace=False))
        rph, cih = rowptr.cpu().numpy(), colind.cpu().numpy()
        sub_ptr = np.zeros(len(rows) + 1, dtype=np.int32)
        sub_ptr[1:] = np.cumsum(rph[rows + 1] - rph[rows])
        sel = np.concatenate([np.arange(rph[r], rph[r + 1]) for r in rows]).astype(np.int64)
        vh = val.cpu().numpy()[sel] if valued else None
        # only the B rows these CSR rows touch travel to the host (B can be tens of GB)
        cols_u, inv = np.unique(cih[sel], return_inverse=True)
        Bsub = B[torch.from_numpy(cols_u.astype(np.int64)).to(dev)].cpu().numpy()
        ref = oracle_py.spmm(sub_ptr, inv.astype(np.int32), vh, Bsub, "fma")
        got = C[torch.from_numpy(rows).to(dev)].cpu().numpy()
        return bool(np.array_equal(got.view(np.uint32), ref.view(np.uint32)))

    # ------------------------------------------------------------------ headline
    N = args.ncols
    res = measure(N, True, args.steps, args.warmup, args.variant)
    if use_dist:  # total non-zeros over all ranks (shards differ in nnz for the RMAT graph)
        tn = torch.tensor([nnz], dtype=torch.int64, device=dev)
        dist.all_reduce(tn)
        nnz_total = int(tn.item())
    else:
        nnz_total = nnz
    flop_per_step = 2.0 * nnz_total * N
    value = flop_per_step * args.steps / res["wall_s"] / 1e9
    ms_per_step = res["wall_s"] / args.steps * 1e3
    abytes = algorithmic_bytes(M, K, N, nnz, True)
    achieved = abytes / (res["kernel_ms"] * 1e-3) / 1e9
    verified = verify(res["B"], res["C"], True) if rank == 0 else None
    roof_gflops = min(2.0 * nnz * N / (abytes / (HBM_PEAK_GBS * 1e9)) / 1e9, FP32_VALU_PEAK_TFLOPS * 1e3)

    traffic = None
    traffic_note = None
    pmc_path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(pmc_path):  # measured with rocprofv3 --pmc (separate passes), see profiles/README.md
        with open(pmc_path) as f:
            pm = json.load(f)
        key = "%s/N%d/valued" % (args.graph, N)
        if key in pm and args.locality == 0.0 and world == 1:
            traffic = pm[key]["bytes_per_launch"]
            traffic_note = pm[key].get("source")

    extra = {}
    if not args.no_extra and world == 1:
        for n2 in widths:
            for valued in (True, False):
                if n2 == N and valued:
                    continue
                torch.cuda.empty_cache()
                need = 4 * (K + M) * n2 * 1.05 + 8 * K * n2  # B + C, plus make_B's int32 temporaries
                if need > torch.cuda.mem_get_info(dev)[0]:
                    extra["N%d_%s" % (n2, "valued" if valued else "unweighted")] = {"skipped": "operands exceed free HBM"}
                    continue
                r2 = measure(n2, valued, max(args.steps // 4, 10), max(args.warmup // 2, 5), args.variant)
                ab = algorithmic_bytes(M, K, n2, nnz, valued)
                extra["N%d_%s" % (n2, "valued" if valued else "unweighted")] = {
                    "gflops": 2.0 * nnz * n2 / (r2["kernel_ms"] * 1e-3) / 1e9,
                    "kernel_us": r2["kernel_ms"] * 1e3,
                    "achieved_GBs": ab / (r2["kernel_ms"] * 1e-3) / 1e9,
                    "frac": ab / (r2["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                }
                del r2

    # BASELINE.json configs[1] names two graphs at N=128: the headline above is the com-Amazon-shaped one, the
    # reddit-shaped one (cache-blocked path) rides along here — a few launches, sampled rows checked.
    if not args.no_extra and world == 1 and args.graph == "com-amazon-like" and args.locality == 0.0:
        torch.cuda.empty_cache()
        g2 = graphs.synthetic_graph("reddit-like", seed=42, device=dev)
        M2, K2, nnz2 = g2["M"], g2["K"], g2["nnz"]
        val2 = torch.rand(nnz2, device=dev) - 0.5
        B2 = make_B(N)[:K2].contiguous() if K2 <= K else (torch.rand(K2, N, device=dev) - 0.5)
        C2 = torch.empty((M2, N), dtype=torch.float32, device=dev)
        for _ in range(2):
            spmm.csr_spmm(g2["rowptr"], g2["colind"], val2, B2, variant=args.variant, out=C2)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            spmm.csr_spmm(g2["rowptr"], g2["colind"], val2, B2, variant=args.variant, out=C2)
        e1.record()
        torch.cuda.synchronize()
        ms2 = e0.elapsed_time(e1) / 5
        plan2 = spmm.SpmmPlan(g2["rowptr"], g2["colind"], K2, N, variant=args.variant)  # split points kept across calls
        spmm.csr_spmm(g2["rowptr"], g2["colind"], val2, B2, variant=args.variant, plan=plan2)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            spmm.csr_spmm(g2["rowptr"], g2["colind"], val2, B2, variant=args.variant, plan=plan2)
        e1.record()
        torch.cuda.synchronize()
        ms2_plan = e0.elapsed_time(e1) / 5
        ab2 = algorithmic_bytes(M2, K2, N, nnz2, True)
        ok2 = verify(B2, C2, True, graph=(g2["rowptr"], g2["colind"], val2, M2))
        extra["reddit-like_N%d_valued" % N] = {
            "gflops": 2.0 * nnz2 * N / (ms2 * 1e-3) / 1e9, "kernel_us": ms2 * 1e3, "nnz": nnz2,
            "achieved_GBs": ab2 / (ms2 * 1e-3) / 1e9, "frac": ab2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "gather_GBs": 4.0 * nnz2 * N / (ms2 * 1e-3) / 1e9, "verified_vs_oracle": ok2,
            "kernel_us_with_plan": ms2_plan * 1e3,
            "note": "launch sequence of the cache-blocked path (split scan + one kernel per column slab)",
        }
        del g2, val2, B2, C2, plan2

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_py

        rph, cih, vh = rowptr.cpu().numpy(), colind.cpu().numpy(), val.cpu().numpy()
        Bh = res["B"].cpu().numpy()
        best = None
        for _ in range(3):  # full pass of the same workload: 2*nnz*N = 0.47 GFLOP, ~0.3 s per pass
            t0 = time.perf_counter()
            oracle_py.spmm(rph, cih, vh, Bh, "golden")
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        t0 = time.perf_counter()
        oracle_py.spmm(rph, cih, vh, Bh, "omp")
        dt_omp = time.perf_counter() - t0
        cpu = {
            "value": 2.0 * nnz * N / best / 1e9,
            "unit": "GFLOP/s",
            "cores": 1,
            "kind": "port",
            "sample": "full %s x N=%d pass (%.2f GFLOP), best of 3, reference loop order i->k->ptr" %
                      (args.graph, N, 2.0 * nnz * N / 1e9),
            "all_cores": {"value": 2.0 * nnz * N / dt_omp / 1e9, "cores": oracle_py.num_threads()},
        }

    if rank == 0:
        out = {
            "metric": "SpMM GFLOP/s (= 2*nnz*N/t), CSR x dense fp32, N=%d" % N,
            "value": value,
            "unit": "GFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": ("RMAT scale %d edge-factor %d (a,b,c,d = .57,.19,.19,.05), %d global nnz, nnz-balanced "
                             "row shards x N=%d, valued CSR, variant %d" %
                             (args.rmat_scale, args.edge_factor, nnz_total, N, args.variant)) if strong else
                            ("%s (M=K=%d, nnz=%d per GPU, symmetric, seed 42+rank, locality %.2f) x N=%d, valued CSR, "
                             "variant %d" % (args.graph, M, nnz, args.locality, N, args.variant)),
                "rows_per_gpu": M,
                "nnz_per_gpu": nnz,
                "ncols": N,
                "partition": ("1-D rows (%s), B replicated by one RCCL broadcast" %
                              ("nnz-balanced shards of one graph" if strong else "one shard per rank"))
                             if world > 1 else "single GPU",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_note,
                "algorithmic_bytes_per_launch": abytes,
                "kernel_us": res["kernel_ms"] * 1e3,
                "roof_gflops": roof_gflops,
                "gflops_kernel": 2.0 * nnz * N / (res["kernel_ms"] * 1e-3) / 1e9,
                # diagnostic: B-row gathers without reuse, 4*nnz*N bytes, and their rate —
                # what the memory system actually moves when B does not fit the L2s
                "gather_bytes_per_launch": 4 * nnz * N,
                "gather_GBs": 4.0 * nnz * N / (res["kernel_ms"] * 1e-3) / 1e9,
            },
            "cpu_baseline": cpu,
            "verified_vs_oracle": verified,
            "exchange_ms": exchange_ms,
            "extra": extra,
        }
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
