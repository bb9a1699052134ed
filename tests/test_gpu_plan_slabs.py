"""Column-slab tables of a plan (round 6; plan.cpp: build_slab_tables, plan_device.hip: device_build_slab_view, spmm_staged.hip: the
continuing launches): a dense clustered matrix is cut into P ascending column ranges, the staged-rows kernel runs once per range and the
second and later launches pick the rows up from the partial sums in C. A partial sum stored and loaded again is the same value and every
row's entries are still added in CSR order, so the bits must equal the oracle's `fma` arithmetic and the plain call's."""
import numpy as np
import pytest
import torch

from helpers import bits

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _community_csr(rng, M, K, comm, deg_in, deg_out, p_empty=0.02, sort=True):
    """Rows in communities of `comm` consecutive ids (then shuffled), ~deg_in columns inside the community's column set, ~deg_out anywhere;
    repeats allowed; columns ascending inside a row when `sort`."""
    rows = []
    ncomm = (M + comm - 1) // comm
    cols_of = [rng.choice(K, size=min(K, 6 * comm), replace=False) for _ in range(ncomm)]
    shuffle = rng.permutation(M)
    for i in range(M):
        c = shuffle[i] // comm
        if rng.rand() < p_empty:
            rows.append(np.zeros(0, dtype=np.int32))
            continue
        a = rng.choice(cols_of[c], size=rng.randint(1, 2 * deg_in))
        b = rng.randint(0, K, size=rng.randint(0, 2 * deg_out + 1))
        r = np.concatenate([a, b]).astype(np.int32)
        rows.append(np.sort(r, kind="stable") if sort else r)
    rowptr = np.zeros(M + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum([len(r) for r in rows])
    return rowptr, np.concatenate(rows).astype(np.int32)


@pytest.mark.parametrize("M,K,slabs", ((3000, 3000, 0), (1111, 5000, 3), (97, 700, 2), (4100, 2500, 16)))
def test_bits_equal_oracle_valued_unweighted_and_new_values(pkg, oracle, monkeypatch, M, K, slabs):
    from gespmm_amd import spmm

    rng = np.random.RandomState(M + slabs)
    rowptr, colind = _community_csr(rng, M, K, comm=150, deg_in=120, deg_out=40)
    nnz = colind.size
    if slabs:
        monkeypatch.setenv("GESPMM_SLABS", str(slabs))  # (read once per process: the first parametrisation decides — kept for direct runs)
    rp, ci = _dev(rowptr), _dev(colind)
    val_h = oracle.hash_val(nnz, seed=11)
    val = _dev(val_h)
    N = 128
    plan = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True, kernel="staged-slabs")
    assert "kernel=staged-slabs" in plan.describe(), plan.describe()
    B_h = oracle.hash_B(K, N, seed=3)
    B = _dev(B_h)
    got = spmm.csr_spmm(rp, ci, val, B, plan=plan).cpu().numpy()
    assert np.array_equal(bits(got), bits(oracle.spmm(rowptr, colind, val_h, B_h, "fma")))
    # C holds garbage (NaN) before the call: the first range's launch must not read it
    C = torch.full((M, N), float("nan"), device="cuda")
    spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan)
    assert np.array_equal(bits(C.cpu().numpy()), bits(got))
    # unweighted through the same plan (the stream carries 1.0f), against the golden loop
    got_u = spmm.csr_spmm_no_edge_value(rp, ci, B, plan=plan).cpu().numpy()
    assert np.array_equal(bits(got_u), bits(oracle.spmm(rowptr, colind, None, B_h, "golden")))
    # other values: re-permuted into the slab view on the device
    val2_h = oracle.hash_val(nnz, seed=12)
    val2 = _dev(val2_h)
    got2 = spmm.csr_spmm(rp, ci, val2, B, plan=plan).cpu().numpy()
    assert np.array_equal(bits(got2), bits(oracle.spmm(rowptr, colind, val2_h, B_h, "fma")))
    # what the slab tables do not serve stays with the plan's other kernels: another width, the max reducer
    B64_h = oracle.hash_B(K, 64, seed=5)
    got64 = spmm.csr_spmm(rp, ci, val2, _dev(B64_h), plan=plan).cpu().numpy()
    assert np.array_equal(bits(got64), bits(oracle.spmm(rowptr, colind, val2_h, B64_h, "fma")))


def test_rows_with_descending_columns_keep_the_other_kernels(pkg, oracle):
    """Range order is CSR order only when every row's columns ascend: otherwise no slab tables are made and the plan's other kernels run."""
    from gespmm_amd import spmm

    rng = np.random.RandomState(5)
    rowptr, colind = _community_csr(rng, 2000, 2000, comm=150, deg_in=100, deg_out=30, sort=False)
    rp, ci = _dev(rowptr), _dev(colind)
    val_h = oracle.hash_val(colind.size, seed=2)
    plan = spmm.SpmmPlan(rp, ci, 2000, 128, values=_dev(val_h), reorder=True, kernel="staged-slabs")
    assert "kernel=staged-slabs" not in plan.describe(), plan.describe()
    B_h = oracle.hash_B(2000, 128, seed=9)
    got = spmm.csr_spmm(rp, ci, _dev(val_h), _dev(B_h), plan=plan).cpu().numpy()
    assert np.array_equal(bits(got), bits(oracle.spmm(rowptr, colind, val_h, B_h, "fma")))


def test_auto_takes_slabs_on_the_dense_community_graph_and_bits_equal_the_plain_call(pkg, oracle):
    """BASELINE configs[1]'s reddit-shaped community stand-in at full size (233 k rows, 115 M entries, mean degree 492): AUTO builds the slab
    tables, the product equals the plain strict-order call bit for bit; sampled rows against the oracle."""
    from gespmm_amd import _lib, graphs, spmm

    g = graphs.synthetic_graph("reddit-sbm", seed=42, device="cuda")
    rp, ci, M, K, nnz = g["rowptr"], g["colind"], g["M"], g["K"], g["nnz"]
    val = torch.rand(nnz, device="cuda") - 0.5
    B = torch.rand(K, 128, device="cuda") - 0.5
    plan = spmm.SpmmPlan(rp, ci, K, 128, values=val)
    assert "kernel=staged-slabs" in plan.describe(), plan.describe()
    got = spmm.csr_spmm(rp, ci, val, B, plan=plan)
    ref = spmm.csr_spmm(rp, ci, val, B, cfg={"flags": _lib.FLAG_STRICT_ORDER})
    assert torch.equal(got.view(torch.int32), ref.view(torch.int32))
    rows = np.random.RandomState(1).choice(M, 64, replace=False)
    rph, cih, vh, Bh = rp.cpu().numpy(), ci.cpu().numpy(), val.cpu().numpy(), B.cpu().numpy()
    sub_ptr = np.zeros(len(rows) + 1, dtype=np.int32)
    sub_ptr[1:] = np.cumsum(rph[rows + 1] - rph[rows])
    sel = np.concatenate([np.arange(rph[r], rph[r + 1]) for r in rows])
    want = oracle.spmm(sub_ptr, cih[sel], vh[sel], Bh, "fma")
    assert np.array_equal(bits(got[torch.from_numpy(rows).cuda()].cpu().numpy()), bits(want))


def test_the_launches_replay_from_a_hip_graph(pkg, oracle):
    """One product = P launches on the caller's stream, nothing allocated: captured once, replayed on new contents of B (C NaN-filled before)."""
    from gespmm_amd import spmm

    rng = np.random.RandomState(21)
    M = K = 2500
    rowptr, colind = _community_csr(rng, M, K, comm=150, deg_in=120, deg_out=40)
    rp, ci = _dev(rowptr), _dev(colind)
    val_h = oracle.hash_val(colind.size, seed=4)
    val = _dev(val_h)
    plan = spmm.SpmmPlan(rp, ci, K, 128, values=val, reorder=True, kernel="staged-slabs")
    assert "kernel=staged-slabs" in plan.describe(), plan.describe()
    B = _dev(oracle.hash_B(K, 128, seed=1))
    C = torch.empty((M, 128), device="cuda")
    fn = lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        fn()
    for seed in (2, 3):
        B_h = oracle.hash_B(K, 128, seed=seed)
        B.copy_(_dev(B_h))
        C.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(bits(C.cpu().numpy()), bits(oracle.spmm(rowptr, colind, val_h, B_h, "fma"))), seed
