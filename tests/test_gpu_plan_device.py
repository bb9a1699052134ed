"""The plan's analysis stage on the device (csrc/plan_device.hip) against its host form (csrc/reorder.cpp, plan.cpp):
the clustering must give the SAME permutation entry by entry (same rules, integer sums), the task tables must be the
same tables, the L2 model must agree with the exact LRU simulation to a point, and — as for every plan — the SpMM
bits do not change."""
import ctypes

import numpy as np
import pytest
import torch

from helpers import bits

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _host_cluster(lib, rp, ci, M, K):
    perm = np.empty(M, dtype=np.int32)
    levels = ctypes.c_int32()
    clusters = (ctypes.c_int32 * 16)()
    rc = lib.gespmm_cluster_rows(rp.ctypes.data, ci.ctypes.data, M, K, 0, perm.ctypes.data, ctypes.byref(levels), clusters)
    assert rc == 0
    return perm, levels.value, list(clusters)


def _device_cluster(lib, rp_d, ci_d, M, K, nnz):
    perm = np.empty(M, dtype=np.int32)
    levels = ctypes.c_int32()
    clusters = (ctypes.c_int32 * 16)()
    rc = lib.gespmm_device_cluster_rows(rp_d.data_ptr(), ci_d.data_ptr(), M, K, nnz, perm.ctypes.data, ctypes.byref(levels),
                                        clusters, None)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return perm, levels.value, list(clusters)


def _graphs(bundled):
    from gespmm_amd import graphs

    out = []
    for g in ("cora", "citeseer", "pubmed"):
        G = bundled[g]
        out.append((g, G["rowptr"], G["colind"], G["M"], G["K"]))
    # planted communities with shuffled ids, a tenth of the com-Amazon stand-in's size; hubs (degree-class kernels 3 and 4)
    gs = graphs.synthetic_graph("com-amazon-sbm", seed=42, device="cpu", scale=0.1)
    out.append(("sbm-0.1", gs["rowptr"].numpy(), gs["colind"].numpy(), gs["M"], gs["K"]))
    rp, ci = graphs.synthetic_csr(30_000, 900_000, symmetric=True, gamma=2.2, seed=5)
    out.append(("power-law", rp.numpy(), ci.numpy(), 30_000, 30_000))
    r = graphs.rmat_shard(15, 16, 0, 1, seed=3, device="cpu")
    out.append(("rmat-15", r["rowptr"].numpy(), r["colind"].numpy(), r["M"], r["K"]))
    # rectangular, empty rows and columns, duplicate entries
    rng = np.random.RandomState(1)
    M, K = 5000, 3000
    deg = rng.randint(0, 9, M)
    deg[rng.rand(M) < 0.2] = 0
    rpx = np.zeros(M + 1, dtype=np.int32)
    rpx[1:] = np.cumsum(deg)
    cix = rng.randint(0, K // 2, int(rpx[-1])).astype(np.int32)
    out.append(("ragged", rpx, cix, M, K))
    return out


def test_device_clustering_gives_the_host_order(pkg, bundled):
    from gespmm_amd import _lib

    lib = _lib.lib
    for name, rp, ci, M, K in _graphs(bundled):
        rp, ci = np.ascontiguousarray(rp, dtype=np.int32), np.ascontiguousarray(ci, dtype=np.int32)
        hp, hl, hc = _host_cluster(lib, rp, ci, M, K)
        dp, dl, dc = _device_cluster(lib, _dev(rp), _dev(ci), M, K, int(rp[-1]))
        assert np.array_equal(np.sort(dp), np.arange(M)), name
        assert (dl, dc) == (hl, hc), (name, dl, dc, hl, hc)
        assert np.array_equal(dp, hp), (name, int((dp != hp).sum()), M)


def test_device_l2_model_tracks_the_lru_simulation(pkg, bundled):
    from gespmm_amd import _lib

    lib = _lib.lib
    for name, rp, ci, M, K in _graphs(bundled):
        rp, ci = np.ascontiguousarray(rp, dtype=np.int32), np.ascontiguousarray(ci, dtype=np.int32)
        nnz = int(rp[-1])
        perm, _, _ = _host_cluster(lib, rp, ci, M, K)
        rp_d, ci_d = _dev(rp), _dev(ci)
        for window in (16, 64, 1024):
            for pm in (None, perm):
                want = lib.gespmm_simulate_l2_hits(rp.ctypes.data, ci.ctypes.data, M, K,
                                                   pm.ctypes.data if pm is not None else None, 8, window)
                got = lib.gespmm_device_l2_model(rp_d.data_ptr(), ci_d.data_ptr(), M, K, nnz,
                                                 pm.ctypes.data if pm is not None else None, 8, window, 0, 8192, None)
                # <= 8192 accesses per slice: every access is evaluated -> exact; otherwise +-1 point (3 sigma of the sample)
                tol = 1e-9 if nnz <= 8 * 8192 else 0.012
                assert abs(got - want) <= tol, (name, window, pm is not None, got, want)


def _tasks(lib, plan, which):
    n = lib.gespmm_plan_debug_tasks(plan._handle, which, None, 0)
    assert n >= 0
    out = np.empty((max(n, 1), 4), dtype=np.int32)
    assert lib.gespmm_plan_debug_tasks(plan._handle, which, out.ctypes.data, n) == n
    return out[:n]


@pytest.mark.parametrize("N", (16, 128))
def test_device_plans_equal_host_plans(pkg, oracle, bundled, N):
    """Same order, same task tables, same model decision (to a point), same SpMM bits."""
    from gespmm_amd import _lib, spmm

    lib = _lib.lib
    for name, rp, ci, M, K in _graphs(bundled):
        rp, ci = np.ascontiguousarray(rp, dtype=np.int32), np.ascontiguousarray(ci, dtype=np.int32)
        nnz = int(rp[-1])
        rp_d, ci_d = _dev(rp), _dev(ci)
        val_h = oracle.hash_val(nnz, seed=3)
        val = _dev(val_h)
        # (strict order: graphs with hub rows would otherwise take the long-row pass, which is tolerance-only)
        pd = spmm.SpmmPlan(rp_d, ci_d, K, N, values=val, reorder=True, analysis="device", flags=_lib.FLAG_STRICT_ORDER)
        ph = spmm.SpmmPlan(rp_d, ci_d, K, N, values=val, reorder=True, analysis="host", flags=_lib.FLAG_STRICT_ORDER)
        assert pd.clustered and ph.clustered, (pd.describe(), ph.describe())
        assert "on the device" in pd.describe() and "on the host" in ph.describe()
        assert np.array_equal(pd.order().numpy(), ph.order().numpy()), name
        for which in (0, 1):
            td, th = _tasks(lib, pd, which), _tasks(lib, ph, which)
            assert td.shape == th.shape and np.array_equal(td, th), (name, which, td.shape, th.shape)
            # a task table covers every row exactly once, in order
            assert td[0, 0] == 0 and np.array_equal(td[1:, 0], td[:-1, 0] + td[:-1, 1]) and td[-1, 0] + td[-1, 1] == M
        B_h = oracle.hash_B(K, N, seed=N)
        B = _dev(B_h)
        a = spmm.csr_spmm(rp_d, ci_d, val, B, plan=pd).cpu().numpy()
        b = spmm.csr_spmm(rp_d, ci_d, val, B, plan=ph).cpu().numpy()
        assert np.array_equal(bits(a), bits(b)), name
        if nnz < 2_000_000:
            assert np.array_equal(bits(a), bits(oracle.spmm(rp, ci, val_h, B_h, "fma"))), name


def test_plan_create_rejects_out_of_range_columns_and_bad_rowptr(pkg, bundled):
    """include/gespmm.h: plans check index ranges (round-2 advice: the host record builders indexed by raw columns)."""
    from gespmm_amd import _lib, spmm

    g = bundled["cora"]
    rp, ci = _dev(g["rowptr"]), _dev(g["colind"])
    bad = ci.clone()
    bad[17] = g["K"]
    for analysis in ("device", "host"):
        with pytest.raises(_lib.GespmmError):
            spmm.SpmmPlan(rp, bad, g["K"], 32, reorder=True, analysis=analysis)
    neg = ci.clone()
    neg[5] = -1
    with pytest.raises(_lib.GespmmError):
        spmm.SpmmPlan(rp, neg, g["K"], 32, reorder=False)
    rp2 = rp.clone()
    rp2[10], rp2[11] = rp[11].item(), rp[10].item() - 1
    with pytest.raises(_lib.GespmmError):
        spmm.SpmmPlan(rp2, ci, g["K"], 32)


def test_full_size_device_plan(pkg):
    """com-Amazon-sized planted communities: the device analysis finds the communities (modelled hits >= 60 %),
    AUTO keeps the clustered order, and the bits equal the plain call's."""
    from gespmm_amd import graphs, spmm

    g = graphs.synthetic_graph("com-amazon-sbm", seed=42, device="cuda")
    val = torch.rand(g["nnz"], device="cuda") - 0.5
    plan = spmm.SpmmPlan(g["rowptr"], g["colind"], g["K"], 128, values=val)
    d = plan.describe()
    assert plan.clustered and "on the device" in d, d
    hits = float(d.split("l2_model=")[1].split()[0].split("->")[1])
    assert hits >= 0.60, d
    B = torch.rand(g["K"], 128, device="cuda") - 0.5
    a = spmm.csr_spmm(g["rowptr"], g["colind"], val, B, plan=plan)
    b = spmm.csr_spmm(g["rowptr"], g["colind"], val, B)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))


def test_release_cached_memory_between_plans(pkg, bundled):
    """The analysis arena is kept for the next plan; gespmm_release_cached_memory gives it back and the next plan still works."""
    from gespmm_amd import _lib, graphs, spmm

    gs = graphs.synthetic_graph("com-amazon-sbm", seed=42, device="cuda", scale=0.1)
    orders = []
    for i in range(3):
        plan = spmm.SpmmPlan(gs["rowptr"], gs["colind"], gs["K"], 64, reorder=True)
        orders.append(plan.order().numpy())
        del plan
        if i == 0:
            free0 = torch.cuda.mem_get_info()[0]
            _lib.lib.gespmm_release_cached_memory()
            assert torch.cuda.mem_get_info()[0] >= free0  # nothing is held any more
    assert np.array_equal(orders[0], orders[1]) and np.array_equal(orders[1], orders[2])
