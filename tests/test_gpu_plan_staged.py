"""The plan's staged-rows kernel (csrc/spmm_staged.hip + plan_device.hip: device_build_staging): one row per wavefront walked
from SGPRs, the most used B rows of every 128-row block read from LDS. Only WHERE a B row comes from changes — every output
element is still one fp32 chain in CSR order — so the bits must equal the oracle's `fma` arithmetic (= the reference's
kernels, tests/test_gpu_ref_kernels.py) and the plain call's."""
import numpy as np
import pytest
import torch

from helpers import bits, edge_case_csr

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("N", (16, 32, 64, 128, 256, 512, 1024))  # 16 / 32 / 64: the lane-group form (spmm_staged_narrow.hip)
@pytest.mark.parametrize("graph", ("cora", "pubmed"))
def test_bits_equal_oracle_valued_unweighted_and_new_values(pkg, oracle, bundled, graph, N):
    from gespmm_amd import spmm

    g = bundled[graph]
    rp, ci = _dev(g["rowptr"]), _dev(g["colind"])
    val_h = oracle.hash_val(g["nnz"], seed=7)
    val = _dev(val_h)
    plan = spmm.SpmmPlan(rp, ci, g["K"], N, values=val, reorder=True, kernel="staged")
    assert plan.clustered and "kernel=staged-rows" in plan.describe(), plan.describe()
    B_h = oracle.hash_B(g["K"], N, seed=N)
    B = _dev(B_h)
    got = spmm.csr_spmm(rp, ci, val, B, plan=plan).cpu().numpy()
    assert np.array_equal(bits(got), bits(oracle.spmm(g["rowptr"], g["colind"], val_h, B_h, "fma"))), (graph, N)
    # the same plan without values: the stream carries 1.0f (fma(1, b, acc) == acc + b), against the golden loop
    got_u = spmm.csr_spmm_no_edge_value(rp, ci, B, plan=plan).cpu().numpy()
    assert np.array_equal(bits(got_u), bits(oracle.spmm(g["rowptr"], g["colind"], None, B_h, "golden"))), (graph, N)
    # with other values (re-permuted and re-interleaved on the device)
    val2_h = oracle.hash_val(g["nnz"], seed=8)
    val2 = _dev(val2_h)
    got2 = spmm.csr_spmm(rp, ci, val2, B, plan=plan).cpu().numpy()
    assert np.array_equal(bits(got2), bits(oracle.spmm(g["rowptr"], g["colind"], val2_h, B_h, "fma")))
    # what the staged kernel does not serve goes to the streaming kernels of the same plan: another width, the max reducer
    B2_h = oracle.hash_B(g["K"], 64, seed=5)
    got64 = spmm.csr_spmm(rp, ci, val2, _dev(B2_h), plan=plan).cpu().numpy()
    assert np.array_equal(bits(got64), bits(oracle.spmm(g["rowptr"], g["colind"], val2_h, B2_h, "fma")))


@pytest.mark.parametrize("N", (128, 256, 512, 1024))
def test_edge_shapes(pkg, oracle, N):
    """Empty rows (also leading / trailing ones inside a block), rows of 1..200 entries, repeated and unsorted columns,
    K != M, M < 128 and M not a multiple of the block size."""
    from gespmm_amd import spmm

    g = edge_case_csr(seed=4)
    rp, ci = _dev(g["rowptr"]), _dev(g["colind"])
    val_h = oracle.hash_val(g["nnz"], seed=3)
    B_h = oracle.hash_B(g["K"], N, seed=N + 1)
    plan = spmm.SpmmPlan(rp, ci, g["K"], N, values=_dev(val_h), reorder=True, kernel="staged")
    assert "kernel=staged-rows" in plan.describe(), plan.describe()
    got = spmm.csr_spmm(rp, ci, _dev(val_h), _dev(B_h), plan=plan).cpu().numpy()
    assert np.array_equal(bits(got), bits(oracle.spmm(g["rowptr"], g["colind"], val_h, B_h, "fma")))
    # several blocks, ragged last block: the same rows tiled 37 times with shifted columns
    reps = 37
    degs = np.tile(np.diff(g["rowptr"]), reps)
    rowptr = np.zeros(degs.size + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(degs)
    colind = np.concatenate([(g["colind"] + 7 * r) % g["K"] for r in range(reps)]).astype(np.int32)
    val_h = oracle.hash_val(colind.size, seed=5)
    rp2, ci2 = _dev(rowptr), _dev(colind)
    plan = spmm.SpmmPlan(rp2, ci2, g["K"], N, values=_dev(val_h), reorder=True, kernel="staged")
    assert "kernel=staged-rows" in plan.describe(), plan.describe()
    got = spmm.csr_spmm(rp2, ci2, _dev(val_h), _dev(B_h), plan=plan).cpu().numpy()
    assert np.array_equal(bits(got), bits(oracle.spmm(rowptr, colind, val_h, B_h, "fma")))


def test_auto_rule_and_full_width_bits_on_a_community_graph(pkg, oracle):
    """products-shaped planted communities at 1/8 size (306 k rows, 15 M entries, mean degree 50): AUTO takes the staged
    kernel, its results equal the plain call's bit for bit (sampled rows also against the oracle); since round 5 the
    com-Amazon-shaped stand-in (mean degree 5.5, 75 % of its entries staged) takes it too, the structureless one (21 %) does not."""
    from gespmm_amd import graphs, spmm

    g = graphs.synthetic_graph("products-sbm", seed=42, device="cuda", scale=0.125)
    rp, ci, M, K, nnz = g["rowptr"], g["colind"], g["M"], g["K"], g["nnz"]
    val = torch.from_numpy(oracle.hash_val(nnz, seed=11)).cuda()
    B = torch.from_numpy(oracle.hash_B(K, 128, seed=12)).cuda()
    plan = spmm.SpmmPlan(rp, ci, K, 128, values=val)
    d = plan.describe()
    assert "kernel=staged-rows" in d, d
    frac = float(d.split("staged_entries=")[1].split()[0])
    assert 0.60 <= frac <= 1.0, d
    got = spmm.csr_spmm(rp, ci, val, B, plan=plan)
    plain = spmm.csr_spmm(rp, ci, val, B, cfg={"flags": 0x100})  # GESPMM_FLAG_STRICT_ORDER: no long-row pass
    assert torch.equal(got.view(torch.int32), plain.view(torch.int32))
    rows = np.random.RandomState(0).choice(M, 400, replace=False)
    rp_h, ci_h, v_h, B_h = rp.cpu().numpy(), ci.cpu().numpy(), val.cpu().numpy(), B.cpu().numpy()
    for r in rows:
        sub_rp = np.array([0, rp_h[r + 1] - rp_h[r]], dtype=np.int32)
        ref = oracle.spmm(sub_rp, ci_h[rp_h[r]:rp_h[r + 1]], v_h[rp_h[r]:rp_h[r + 1]], B_h, "fma")
        assert np.array_equal(bits(got[r:r + 1].cpu().numpy()), bits(ref)), r
    # the same tables twice: the analysis is deterministic
    plan2 = spmm.SpmmPlan(rp, ci, K, 128, values=val)
    assert plan2.describe().split("staged_entries=")[1].split()[0] == d.split("staged_entries=")[1].split()[0]
    # operands that are not 16-byte aligned fall back to the streaming kernels of the same plan
    Boff = torch.empty(K * 128 + 1, device="cuda")[1:].view(K, 128)
    Boff.copy_(B)
    assert torch.equal(spmm.csr_spmm(rp, ci, val, Boff, plan=plan).view(torch.int32), plain.view(torch.int32))
    del plan, plan2

    g = graphs.synthetic_graph("com-amazon-sbm", seed=42, device="cuda")
    plan = spmm.SpmmPlan(g["rowptr"], g["colind"], g["K"], 128)
    assert plan.clustered and "kernel=staged-rows" in plan.describe(), plan.describe()
    g = graphs.synthetic_graph("com-amazon-like", seed=42, device="cuda")
    plan = spmm.SpmmPlan(g["rowptr"], g["colind"], g["K"], 128, expected_launches=5000)  # (enough launches for the analysis to run at all)
    assert plan.clustered and "kernel=staged-rows" not in plan.describe(), plan.describe()


def test_hub_rows_are_handed_to_the_long_row_pass(pkg, oracle):
    """Rows beyond 2048 entries would be one wavefront's serial walk: the staged kernel sees them EMPTY and the streaming kernel
    gets them as one-row tasks. With GESPMM_FLAG_STRICT_ORDER every row is still one strict chain (bit-exact); without it the hubs
    take the long-row pass — a re-association held to 1e-4 * sum|a.b| — and every other row keeps its bits."""
    from gespmm_amd import spmm

    rng = np.random.RandomState(3)
    M = K = 4000
    degs = rng.randint(1, 30, size=M)
    hubs = {17: 3000, 128: 2049, 2500: 9000, M - 1: 2600}
    for r, d in hubs.items():
        degs[r] = d
    degs[300] = 2048  # at the limit: stays with the staged kernel
    rowptr = np.zeros(M + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(degs)
    colind = rng.randint(0, K, size=int(rowptr[-1])).astype(np.int32)
    val_h = oracle.hash_val(colind.size, seed=2)
    rp, ci, val = _dev(rowptr), _dev(colind), _dev(val_h)
    for N in (128, 256, 512):
        B_h = oracle.hash_B(K, N, seed=3 + N)
        B = _dev(B_h)
        ref = oracle.spmm(rowptr, colind, val_h, B_h, "fma")
        plan = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True, kernel="staged", flags=0x100)  # STRICT_ORDER
        assert "kernel=staged-rows" in plan.describe() and "hub_rows=4" in plan.describe(), plan.describe()
        got = spmm.csr_spmm(rp, ci, val, B, plan=plan).cpu().numpy()
        assert np.array_equal(bits(got), bits(ref)), N
        got_u = spmm.csr_spmm_no_edge_value(rp, ci, B, plan=plan).cpu().numpy()
        assert np.array_equal(bits(got_u), bits(oracle.spmm(rowptr, colind, None, B_h, "golden"))), N
        got2 = spmm.csr_spmm(rp, ci, val, B, plan=plan).cpu().numpy()  # values back in (interleaved again through the split map)
        assert np.array_equal(bits(got2), bits(ref)), N
        # without the flag: hubs through the long-row pass
        plan = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True, kernel="staged", flags=0x200)  # SPLIT_LONG_ROWS (small matrix)
        assert "hub_rows=4" in plan.describe(), plan.describe()
        got = spmm.csr_spmm(rp, ci, val, B, plan=plan).cpu().numpy()
        others = np.ones(M, dtype=bool)
        others[list(hubs)] = False
        assert np.array_equal(bits(got[others]), bits(ref[others])), N
        for r in hubs:
            lo, hi = rowptr[r], rowptr[r + 1]
            bound = 1e-4 * (np.abs(val_h[lo:hi, None].astype(np.float64) * B_h[colind[lo:hi]].astype(np.float64))).sum(0)
            assert np.all(np.abs(got[r].astype(np.float64) - ref[r].astype(np.float64)) <= bound + 1e-30), (N, r)


def _random_local_csr(rng, M, K, max_deg, local, p_empty):
    """Rows whose columns come mostly from a window around the row (shared inside a block), the rest from anywhere;
    repeated and unsorted columns, empty rows."""
    degs = rng.randint(1, max_deg + 1, size=M)
    degs[rng.rand(M) < p_empty] = 0
    rowptr = np.zeros(M + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(degs)
    rows = np.repeat(np.arange(M), degs)
    near = (rows * K // max(M, 1) + rng.randint(-local, local + 1, size=rows.size)) % K
    far = rng.randint(0, K, size=rows.size)
    colind = np.where(rng.rand(rows.size) < 0.75, near, far).astype(np.int32)
    return rowptr, colind


@pytest.mark.parametrize("seed", range(12))
def test_random_matrices_equal_the_plain_call(pkg, oracle, seed):
    """Seeded random matrices, square (far columns marked) and rectangular, 1 .. 6000 rows, rows up to the 2048-entry limit:
    the staged kernel's bits = the plain call's strict-order bits (which the other tests pin to the oracle)."""
    from gespmm_amd import spmm

    rng = np.random.RandomState(1000 + seed)
    M = int(rng.choice([1, 2, 127, 128, 129, 700, 3000, 6000]))
    K = M if seed % 3 else int(rng.randint(1, 5000))
    max_deg = int(rng.choice([3, 40, 300]))
    rowptr, colind = _random_local_csr(rng, M, K, max_deg, local=int(rng.choice([2, 30, 400])), p_empty=float(rng.choice([0.0, 0.3])))
    if seed % 4 == 1 and M >= 128:  # one row at the limit
        extra = rng.randint(0, K, size=2048).astype(np.int32)
        r = int(rng.randint(0, M))
        colind = np.concatenate([colind[:rowptr[r]], extra, colind[rowptr[r + 1]:]])
        d = 2048 - (rowptr[r + 1] - rowptr[r])
        rowptr = rowptr.copy()
        rowptr[r + 1:] += d
    if colind.size == 0:
        colind = np.zeros(1, dtype=np.int32)
        rowptr[-1] = 1 if M == 1 else rowptr[-1]
        rowptr[1:] = np.maximum(rowptr[1:], 0)
        rowptr[M] = 1
    rp, ci = _dev(rowptr), _dev(colind)
    val = _dev(oracle.hash_val(colind.size, seed=seed))
    for N in (128, 256, 512) + ((1024,) if seed % 4 == 0 else ()):  # (beyond 256 columns: 256-column tiles bound to XCDs)
        B = _dev(oracle.hash_B(K, N, seed=seed + N))
        plan = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True, kernel="staged", flags=0x100)
        if plan.clustered:  # (matrices of a few rows keep their storage order: nothing to stage)
            assert "kernel=staged-rows" in plan.describe(), plan.describe()
        want = spmm.csr_spmm(rp, ci, val, B, cfg={"flags": 0x100})
        got = spmm.csr_spmm(rp, ci, val, B, plan=plan)
        assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (seed, M, K, N, max_deg)
        want_u = spmm.csr_spmm_no_edge_value(rp, ci, B, cfg={"flags": 0x100})
        got_u = spmm.csr_spmm_no_edge_value(rp, ci, B, plan=plan)
        assert torch.equal(got_u.view(torch.int32), want_u.view(torch.int32)), (seed, M, K, N, "unweighted")


def test_dense_graphs_are_clustered_only_with_strong_communities(pkg, oracle):
    """Dense graphs take the cache-blocked path; AUTO plans now run the analysis on them too and keep the clustered order only when
    the L2 model promises >= 65 % hits (a reddit-shaped graph WITH planted communities: 3.0 vs 4.0 ms at full size); the structureless
    stand-in keeps the cache-blocked path. Same bits either way."""
    from gespmm_amd import graphs, spmm

    M, nnz = 58240, 28_653_972
    rp, ci, _ = graphs.community_csr(M, nnz, 72, 8, 330.0, 0.6, 1.5, 1.55, 42, "cuda")
    val = torch.from_numpy(oracle.hash_val(int(ci.numel()), seed=21)).cuda()
    for N in (64, 128, 256):
        B = torch.from_numpy(oracle.hash_B(M, N, seed=N)).cuda()
        plan = spmm.SpmmPlan(rp, ci, M, N, values=val)
        d = plan.describe()
        assert d.startswith("order=clustered") and "slab-blocked" not in d, d
        after = float(d.split("l2_model=")[1].split()[0].split("->")[1])
        assert after >= 0.65, d
        got = spmm.csr_spmm(rp, ci, val, B, plan=plan)
        plain = spmm.csr_spmm(rp, ci, val, B, cfg={"flags": 0x100})
        assert torch.equal(got.view(torch.int32), plain.view(torch.int32)), N
        del plan
    g = graphs.synthetic_graph("reddit-like", seed=42, device="cuda", scale=0.25)
    plan = spmm.SpmmPlan(g["rowptr"], g["colind"], g["K"], 128)
    d = plan.describe()
    assert d.startswith("order=storage") and "kernel=slab-blocked" in d, d
    B = torch.from_numpy(oracle.hash_B(g["K"], 128, seed=9)).cuda()
    got = spmm.csr_spmm_no_edge_value(g["rowptr"], g["colind"], B, plan=plan)
    plain = spmm.csr_spmm_no_edge_value(g["rowptr"], g["colind"], B, cfg={"flags": 0x100})
    assert torch.equal(got.view(torch.int32), plain.view(torch.int32))


def test_b_beyond_4gb_takes_the_two_halves_base(pkg, oracle):
    """N = 512 with K * N * 4 between 4 and 8 GB (products-shaped: 5.0 GB): the lane's 32-bit offset wraps modulo 4 GB and the base
    pointer is chosen between B and B + 4 GB by a bit of the scalar code. Columns on both sides of the boundary, and within a row of
    it; staged and gathered entries on both sides; bits = the plain call's strict-order bits."""
    from gespmm_amd import spmm

    N = 512
    K = 2_300_000  # 4.71 GB of B; the 4 GB boundary is at row 2 097 152
    M = 40_000
    boundary = (1 << 32) // (N * 4)
    free, _ = torch.cuda.mem_get_info()
    if free < 12 * (1 << 30):
        pytest.skip("needs ~6 GB of device memory")
    rng = np.random.RandomState(5)
    degs = rng.randint(4, 40, size=M)
    rowptr = np.zeros(M + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(degs)
    rows = np.repeat(np.arange(M), degs)
    # three kinds of columns: a window that follows the row (reused inside a block: staged), spread over BOTH halves; uniform over K;
    # and the rows right at the boundary
    near = (boundary - 20_000 + rows + rng.randint(-40, 41, size=rows.size)) % K
    anywhere = rng.randint(0, K, size=rows.size)
    edge = boundary + rng.randint(-2, 3, size=rows.size)
    pick = rng.rand(rows.size)
    colind = np.where(pick < 0.6, near, np.where(pick < 0.95, anywhere, edge)).astype(np.int32)
    assert (colind < boundary).any() and (colind >= boundary).any()
    rp, ci = _dev(rowptr), _dev(colind)
    val = _dev(oracle.hash_val(colind.size, seed=9))
    B = torch.empty((K, N), dtype=torch.float32, device="cuda")
    step = 1 << 18
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    for r0 in range(0, K, step):  # values that differ between the halves at the same offset (a wrong base would show)
        r1 = min(K, r0 + step)
        B[r0:r1] = (torch.randint(0, 100, (r1 - r0, N), generator=g, device="cuda", dtype=torch.int32) - 50).float() / 100
    plan = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True, kernel="staged", flags=0x100)
    d = plan.describe()
    assert "kernel=staged-rows" in d, d
    got = spmm.csr_spmm(rp, ci, val, B, plan=plan)
    want = spmm.csr_spmm(rp, ci, val, B, cfg={"flags": 0x100})
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    # sampled rows against the oracle too (the plain call above takes 64-bit offsets: another code path, same bits)
    rows_s = rng.choice(M, 64, replace=False)
    v_h, B_rows = val.cpu().numpy(), None
    for r in rows_s:
        lo, hi = rowptr[r], rowptr[r + 1]
        cols = colind[lo:hi]
        Bs = B[torch.from_numpy(cols.astype(np.int64)).cuda()].cpu().numpy()
        ref = oracle.spmm(np.array([0, hi - lo], dtype=np.int32), np.arange(hi - lo, dtype=np.int32), v_h[lo:hi], Bs, "fma")
        assert np.array_equal(bits(got[r:r + 1].cpu().numpy()), bits(ref)), r
    # beyond 8 GB the staged kernel is not offered: the plan falls back to the streaming kernels
    assert pkg._lib.plan_policy(M, 4_300_000, colind.size, 512, 40, 0.0, 0.9, 0.9, kernel=pkg._lib.PLAN_KERNEL_STAGED)["build_staged"] == 0


def test_reduced_soak_of_the_staged_kernel(pkg):
    """scripts/staged_soak.py at a reduced count in the GPU suite (round-4 review: the inline-assembly kernel is soak-tested, not proven —
    keep the soak running with every round, not only as a log): 200 seeded random matrices x N = 128 / 256 / 512 (/ 1024), valued and
    unweighted, forced staged plans against the plain call's strict-order bits, gespmm_plan_tune on every fifth seed."""
    import os
    import sys

    from helpers import ROOT

    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import staged_soak

    checked, staged = staged_soak.soak(91000, 200, verbose=False)
    assert checked >= 1200 and staged >= 300, (checked, staged)
