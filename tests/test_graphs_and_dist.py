"""Synthetic stand-in generators (host logic) and the multi-GPU row partition +
B exchange, exercised with 2 gloo processes on CPU tensors. The SpMM of each shard is
evaluated by the ORACLE here (as the checker): what is under test is the partition,
the rebasing and the exchange, which contain no device code."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT


def test_synthetic_contract_small(pkg):
    from gespmm_amd import graphs

    for name in ("com-amazon-like", "cit-hepth-like", "pubmed-like"):
        g = graphs.synthetic_graph(name, seed=3, scale=0.02)
        M, nnz, sym, _ = graphs.SPECS[name]
        rp, ci = g["rowptr"].long(), g["colind"].long()
        assert rp[0] == 0 and rp[-1] == g["nnz"] == ci.numel()
        rows = torch.repeat_interleave(torch.arange(g["M"]), rp[1:] - rp[:-1])
        key = rows * g["M"] + ci
        assert torch.all(key[1:] > key[:-1]), "sorted by (row, col), no duplicates"
        assert int(ci.min()) >= 0 and int(ci.max()) < g["K"]
        if sym:
            assert not torch.any(rows == ci)
            assert torch.equal(torch.sort(ci * g["M"] + rows)[0], key)
        g2 = graphs.synthetic_graph(name, seed=3, scale=0.02)
        assert torch.equal(g2["colind"], g["colind"]), "pure function of (name, seed)"
        g3 = graphs.synthetic_graph(name, seed=4, scale=0.02)
        assert not torch.equal(g3["colind"], g["colind"])


def test_exact_sizes_of_the_headline_graph(pkg):
    from gespmm_amd import graphs

    g = graphs.synthetic_graph("cit-hepth-like", seed=42)
    assert (g["M"], g["nnz"]) == (27770, 352807)  # matrix_id_info.xlsx row cit-HepTh
    loc = graphs.synthetic_graph("com-amazon-like", seed=1, scale=0.05, locality=0.9, band=50)
    rp, ci = loc["rowptr"].long(), loc["colind"].long()
    rows = torch.repeat_interleave(torch.arange(loc["M"]), rp[1:] - rp[:-1])
    d = (rows - ci).abs()
    d = torch.minimum(d, loc["M"] - d)
    assert float((d < 500).float().mean()) > 0.8, "locality knob produces a banded pattern"


def test_self_loops_and_transpose(pkg):
    from gespmm_amd import graphs
    import scipy.sparse as sp

    g = graphs.synthetic_graph("pubmed-selfloop-like", seed=0, scale=0.05)
    M = g["M"]
    A = sp.csr_matrix((np.ones(g["nnz"]), g["colind"].numpy(), g["rowptr"].numpy()), shape=(M, M))
    assert np.all(A.diagonal() == 1)
    colptr, rowind = graphs.transpose_csr(g["rowptr"], g["colind"])
    T = A.tocsc()
    assert np.array_equal(colptr.numpy(), T.indptr) and np.array_equal(rowind.numpy(), T.indices)
    B = graphs.reference_B(7, 5, seed=1)
    q = torch.round(B * 100)
    assert q.min() >= -50 and q.max() <= 49


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gespmm_amd  # noqa: F401
    from gespmm_amd import dist as gdist
    from gespmm_amd import graphs

    import oracle_py

    G = graphs.load_mtx_as_csr(os.path.join(ROOT, "tests", "golden", "cora.mtx"))
    N = 24
    val = oracle_py.hash_val(G["nnz"], seed=1)
    cut = gdist.partition_rows(G["rowptr"], world)
    lptr, lcol, lval, (r0, r1) = gdist.shard_csr(G["rowptr"], G["colind"], val, cut, rank)
    assert lptr[0] == 0 and lptr[-1] == len(lcol) == len(lval)

    # (a) B row-sharded over ranks (ragged shards), all_gather'ed
    Bfull_ref = oracle_py.hash_B(G["K"], N, seed=2)
    bcut = np.linspace(0, G["K"], world + 1).astype(int)
    bcut[1] += 3  # ragged on purpose
    counts = [int(bcut[i + 1] - bcut[i]) for i in range(world)]
    mine = torch.from_numpy(Bfull_ref[bcut[rank]:bcut[rank + 1]].copy())
    Bfull = gdist.exchange_dense(mine, counts)
    assert np.array_equal(Bfull.numpy(), Bfull_ref)
    # equal shards take the all_gather_into_tensor path
    Keq = (G["K"] // world) * world
    eq = torch.from_numpy(Bfull_ref[rank * (Keq // world):(rank + 1) * (Keq // world)].copy())
    assert np.array_equal(gdist.exchange_dense(eq).numpy(), Bfull_ref[:Keq])
    # (b) B owned by rank 1, broadcast
    Bb = gdist.broadcast_dense(torch.from_numpy(Bfull_ref) if rank == 1 else None, G["K"], N, src=1)
    assert np.array_equal(Bb.numpy(), Bfull_ref)

    # (c) the column-panel pipeline (exchange inside the product): each rank owns K/world rows of B, panels travel by
    #     all-gather, the shard product of every panel (oracle as checker on the host) == the columns of the full product
    pcounts = [(G["K"] * (r + 1)) // world - (G["K"] * r) // world for r in range(world)]
    kk0 = sum(pcounts[:rank])
    mine_rows = torch.from_numpy(Bfull_ref[kk0:kk0 + pcounts[rank]].copy())
    widths = [8, 8, 5, 3]
    assert sum(widths) == N
    cols = np.concatenate([[0], np.cumsum(widths)])
    gdist._MAX_ELEMS = 5000  # force the row-chunked staging path of the all-gather (a real run chunks at 2^31 elements)

    def host_product(rowptr_, colind_, val_, Bp, out):
        out.copy_(torch.from_numpy(oracle_py.spmm(rowptr_.numpy(), colind_.numpy(), val_.numpy(), Bp.numpy(), "fma")))
        return out

    pipe = gdist.PanelPipeline(torch.from_numpy(lptr), torch.from_numpy(lcol), torch.from_numpy(lval), G["K"], pcounts,
                               widths, "cpu", product=host_product)
    Cp = pipe.run([mine_rows[:, cols[i]:cols[i + 1]].contiguous() for i in range(len(widths))])
    C_cols = oracle_py.spmm(lptr, lcol, lval, Bfull_ref, "fma")
    for i in range(len(widths)):
        assert np.array_equal(Cp[i].numpy().view(np.uint32), C_cols[:, cols[i]:cols[i + 1]].copy().view(np.uint32)), i
    gdist._MAX_ELEMS = (1 << 31) - 1024

    # shard product (oracle as checker) == the same rows of the full product
    C_loc = oracle_py.spmm(lptr, lcol, lval, Bfull.numpy(), "fma")
    C_ref = oracle_py.spmm(G["rowptr"], G["colind"], val, Bfull_ref, "fma")
    assert np.array_equal(C_loc.view(np.uint32), C_ref[r0:r1].view(np.uint32))
    # gather the row shards back and compare the whole matrix on rank 0
    pieces = [None] * world
    dist.all_gather_object(pieces, (r0, r1, C_loc))
    if rank == 0:
        whole = np.concatenate([p[2] for p in sorted(pieces, key=lambda t: t[0])])
        assert np.array_equal(whole.view(np.uint32), C_ref.view(np.uint32))
        loads = [int(G["rowptr"][p[1]] - G["rowptr"][p[0]]) for p in pieces]
        assert max(loads) - min(loads) <= 2 * 168 + 2, "nnz-balanced (cora max degree 168)"
        open(os.path.join(tmpdir, "ok"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_row_partition_and_exchange_world2(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def test_rmat_shards_tile_the_global_graph(pkg):
    from gespmm_amd import graphs

    full = graphs.rmat_shard(11, 8, 0, 1, seed=5, chunk_edges=3000)
    assert full["M"] == full["K"] == 2048 and full["nnz"] == full["global_nnz"] == 8 * 2048
    for balanced in (False, True):
        parts = [graphs.rmat_shard(11, 8, r, 3, seed=5, chunk_edges=3000, balanced=balanced) for r in range(3)]
        assert parts[0]["row_begin"] == 0 and parts[-1]["row_end"] == 2048
        assert all(parts[i]["row_end"] == parts[i + 1]["row_begin"] for i in range(2))
        assert torch.equal(torch.cat([p["colind"] for p in parts]), full["colind"])
        offs = 0
        for p in parts:
            seg = full["rowptr"][p["row_begin"]:p["row_end"] + 1] - full["rowptr"][p["row_begin"]]
            assert torch.equal(seg, p["rowptr"])
            offs += p["nnz"]
        assert offs == full["nnz"]
        if balanced:  # same cuts as the C ABI partitioner on the global rowptr
            cut = graphs.row_partition(full["rowptr"].numpy(), 3)
            assert list(cut) == parts[0]["cuts"]
    deg = (full["rowptr"][1:] - full["rowptr"][:-1])
    assert int(deg.max()) > 20 * float(deg.float().mean()), "RMAT is heavy-tailed"


def _bench_rmat_worker(rank, world, port, tmpdir):
    """bench.py's several-GPU mode (run_rmat: rank-local RMAT shards, B exchange, kernel-only loop, panel pipeline, the record)
    under gloo on the host: the checker is INJECTED as the product (tests only — bench.py never supplies a host product)."""
    import argparse
    import json

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import bench
    import oracle_py

    from gespmm_amd import graphs

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def host_product(rowptr_, colind_, val_, B_, out):
        out.copy_(torch.from_numpy(oracle_py.spmm(rowptr_.numpy(), colind_.numpy(), val_.numpy(), B_.contiguous().numpy(), "fma")))
        return out

    scale, ef, N = 12, 16, 20
    whole_ok = []

    def on_local_product(g, val, B, C):
        pieces = [None] * world
        dist.all_gather_object(pieces, (g["row_begin"], g["row_end"], val.numpy(), C.numpy().copy()))
        if rank != 0:
            return
        pieces.sort(key=lambda t: t[0])
        assert pieces[0][0] == 0 and pieces[-1][1] == 1 << scale
        assert all(pieces[i][1] == pieces[i + 1][0] for i in range(world - 1)), "contiguous row shards"
        full = graphs.rmat_shard(scale, ef, 0, 1, seed=42)
        vals = np.concatenate([p[2] for p in pieces])
        assert vals.shape[0] == full["nnz"] == ef << scale
        ref = oracle_py.spmm(full["rowptr"].numpy(), full["colind"].numpy(), vals, B.numpy(), "fma")
        got = np.concatenate([p[3] for p in pieces])
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "shard rows concatenate to the unsharded product"
        whole_ok.append(True)

    env = bench.BenchEnv(torch, dist, "cpu", world, rank, True, product=host_product, on_local_product=on_local_product)
    args = argparse.Namespace(rmat_scale=scale, edge_factor=ef, steps=2, warmup=1, panel_cols=8, variant=-1)
    out = bench.run_rmat(args, env, N)
    if rank == 0:
        assert whole_ok == [True]
        line = json.loads(bench.compact_line(out))
        assert line["n_gpus"] == world and line["scaling"] == "strong" and line["value"] > 0
        assert line["config"]["nnz_per_gpu"] > 0 and "RMAT scale 12" in line["config"]["workload"]
        e2e = line["exchange"]["end_to_end"]
        assert e2e["panels"] == 3 and e2e["panel_cols"] == 8, e2e
        assert e2e["sampled_rows_bit_equal_resident_product"] == "%d of %d" % (3 * out["config"]["rows_per_gpu"], 3 * out["config"]["rows_per_gpu"])
        assert line["exchange"]["bytes_received_per_gpu"] > 0 and line["exchange"]["amortised_over_L"]["L"] == 64
        assert "one_gpu_reference" in line and "rmat-12_N256" in line["one_gpu_reference"]
        assert line["verified_vs_oracle"]["failed"] == 0
        assert line["roofline"]["frac"] > 0 and line["cpu_baseline"] is None
        open(os.path.join(tmpdir, "ok%d" % world), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])  # 2: equal B shards (all-gather); 3: ragged shards (one broadcast per owner)
def test_bench_rmat_mode_under_gloo(tmp_path, world, pkg):
    port = 31500 + (os.getpid() % 2000) + world
    mp.spawn(_bench_rmat_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / ("ok%d" % world)).exists()
