"""gespmm_plan (row-clustered copy + nnz-balanced task table): the processing order is the ONLY thing a plan
changes, so every result must have the same bits as the plain call and as the oracle."""
import numpy as np
import pytest
import torch

from helpers import bits, edge_case_csr, sampled_rows_equal_oracle

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("graph", ("cora", "pubmed"))
def test_clustered_plan_bits_equal_oracle(pkg, oracle, bundled, graph):
    from gespmm_amd import spmm

    g = bundled[graph]
    rp, ci = _dev(g["rowptr"]), _dev(g["colind"])
    val_h = oracle.hash_val(g["nnz"], seed=7)
    val = _dev(val_h)
    for N, kernel in ((3, "auto"), (32, "stream"), (128, "stream"), (128, "seg-stream"), (32, "seg-stream"), (100, "seg-stream"),
                      (64, "stream"), (256, "seg-stream"), (512, "stream"), (130, "stream"), (101, "stream"), (260, "seg-stream"),
                      (512, "auto"), (16, "stream"), (8, "seg-stream")):
        plan = spmm.SpmmPlan(rp, ci, g["K"], N, values=val, reorder=True, kernel=kernel)
        assert plan.clustered, plan.describe()
        order = plan.order().numpy()
        assert np.array_equal(np.sort(order), np.arange(g["M"]))
        B_h = oracle.hash_B(g["K"], N, seed=N)
        B = _dev(B_h)
        got = spmm.csr_spmm(rp, ci, val, B, plan=plan).cpu().numpy()
        ref = oracle.spmm(g["rowptr"], g["colind"], val_h, B_h, "fma")
        assert np.array_equal(bits(got), bits(ref)), (graph, N)
        # the same plan without values = the unweighted kernels, against the golden loop
        got_u = spmm.csr_spmm_no_edge_value(rp, ci, B, plan=plan).cpu().numpy()
        ref_u = oracle.spmm(g["rowptr"], g["colind"], None, B_h, "golden")
        assert np.array_equal(bits(got_u), bits(ref_u)), (graph, N)
        # and with values again (re-permuted on the device)
        val2_h = oracle.hash_val(g["nnz"], seed=8)
        val2 = _dev(val2_h)
        got2 = spmm.csr_spmm(rp, ci, val2, B, plan=plan).cpu().numpy()
        assert np.array_equal(bits(got2), bits(oracle.spmm(g["rowptr"], g["colind"], val2_h, B_h, "fma")))
        del plan


def test_values_edited_in_place_are_picked_up(pkg, oracle, bundled):
    from gespmm_amd import spmm

    g = bundled["citeseer"]
    rp, ci = _dev(g["rowptr"]), _dev(g["colind"])
    val = _dev(oracle.hash_val(g["nnz"], seed=1))
    B = _dev(oracle.hash_B(g["K"], 64, seed=2))
    plan = spmm.SpmmPlan(rp, ci, g["K"], 64, values=val, reorder=True)
    a = spmm.csr_spmm(rp, ci, val, B, plan=plan)
    assert torch.equal(a, spmm.csr_spmm(rp, ci, val, B))
    val.mul_(-2.0)  # same tensor, new contents
    b = spmm.csr_spmm(rp, ci, val, B, plan=plan)
    assert torch.equal(b, spmm.csr_spmm(rp, ci, val, B)) and not torch.equal(a, b)
    rp[1:3] = rp[1:3]  # an in-place write to the pattern (even a no-op) invalidates the plan
    with pytest.raises(ValueError):
        spmm.csr_spmm(rp, ci, val, B, plan=plan)


def test_edge_shapes_rectangular_empty_rows_duplicates(pkg, oracle):
    from gespmm_amd import spmm

    g = edge_case_csr(seed=4)
    rp, ci = _dev(g["rowptr"]), _dev(g["colind"])
    val_h = oracle.hash_val(g["nnz"], seed=3)
    for N in (1, 7, 64, 132):
        B_h = oracle.hash_B(g["K"], N, seed=N + 1)
        for te, kernel in ((0, "stream"), (8, "stream"), (1000, "stream"), (0, "seg-stream"), (8, "seg-stream"), (1000, "seg-stream")):
            plan = spmm.SpmmPlan(rp, ci, g["K"], N, values=_dev(val_h), reorder=True, task_entries=te, kernel=kernel)
            got = plan.run(None, _dev(B_h)).cpu().numpy()
            ref = oracle.spmm(g["rowptr"], g["colind"], val_h, B_h, "fma")
            assert np.array_equal(bits(got), bits(ref)), (N, te)
            mx = spmm.SpmmPlan(rp, ci, g["K"], N, reorder=True, task_entries=te, kernel=kernel).run(None, _dev(B_h), reduce_max=-10000.0)
            assert np.array_equal(bits(mx.cpu().numpy()), bits(oracle.spmm_max(g["rowptr"], g["colind"], B_h)))


def test_fuzz_plans_against_plain_calls(pkg):
    from gespmm_amd import _lib, spmm

    rng = np.random.RandomState(5)
    for case in range(60):
        M = int(rng.randint(1, 3000))
        K = int(rng.randint(1, 3000))
        deg = rng.geometric(0.2, size=M) - 1
        if case % 5 == 0:
            deg[rng.randint(0, M)] = rng.randint(100, 5000)
        rowptr = np.zeros(M + 1, dtype=np.int32)
        rowptr[1:] = np.cumsum(deg)
        colind = rng.randint(0, K, size=int(rowptr[-1])).astype(np.int32)
        N = int(rng.choice([1, 2, 5, 16, 31, 32, 33, 64, 96, 128, 129, 256, 300]))
        rp, ci = _dev(rowptr), _dev(colind)
        val = torch.rand(colind.size, device="cuda") - 0.5
        B = torch.rand(K, N, device="cuda") - 0.5
        # hub rows: the plan would switch the long-row pass on (a re-association); pin both sides to the strict chain
        # (also through the 64-bit-offset, non-temporal-store, 4-deep-unroll and no-XCD-remap instantiations of the planned kernels)
        extra = int(rng.choice([0, _lib.FLAG_FORCE_IDX64, _lib.FLAG_NT_STORE, _lib.FLAG_SC1_STORE, _lib.FLAG_SHALLOW_UNROLL, _lib.FLAG_NO_XCD_REMAP,
                                _lib.FLAG_FORCE_IDX64 | _lib.FLAG_SHALLOW_UNROLL]))
        plan = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True, task_entries=int(rng.choice([0, 16, 64])),
                             flags=_lib.FLAG_STRICT_ORDER | extra,
                             kernel=str(rng.choice(["auto", "stream", "seg-stream"])))
        got = spmm.csr_spmm(rp, ci, val, B, plan=plan)
        ref = spmm.csr_spmm(rp, ci, val, B, cfg={"flags": _lib.FLAG_STRICT_ORDER})
        assert torch.equal(got.view(torch.int32), ref.view(torch.int32)), (case, M, K, N)


def test_full_size_community_graph_auto_plan(pkg, oracle):
    """com-Amazon-sized planted-community graph with shuffled ids: AUTO clusters it (the L2 model predicts the
    gain), bits equal the plain call on the whole matrix and the oracle on sampled rows."""
    from gespmm_amd import graphs, spmm

    g = graphs.synthetic_graph("com-amazon-sbm", seed=42, device="cuda")
    assert g["M"] == 334863 and g["nnz"] == 1851744
    rp, ci, M = g["rowptr"], g["colind"], g["M"]
    val = torch.rand(g["nnz"], device="cuda") - 0.5
    plan = spmm.SpmmPlan(rp, ci, M, 128, values=val)
    d = plan.describe()
    assert d.startswith("order=clustered"), d
    assert "kernel=staged-rows" in d and "group_tasks=" in d, d  # round 5: the staged-rows kernel on short rows too (92.7 vs 107 us)
    B = (torch.randint(0, 100, (M, 128), device="cuda", dtype=torch.int32) - 50).float() / 100
    got = spmm.csr_spmm(rp, ci, val, B, plan=plan)
    ref = spmm.csr_spmm(rp, ci, val, B)
    assert torch.equal(got.view(torch.int32), ref.view(torch.int32))
    assert sampled_rows_equal_oracle(oracle, rp, ci, val, B, got, nrows=1024), "planned headline product differs from the oracle"
    # structureless stand-in: whatever AUTO decides, the bits stay
    g2 = graphs.synthetic_graph("com-amazon-like", seed=42, device="cuda")
    plan2 = spmm.SpmmPlan(g2["rowptr"], g2["colind"], M, 128, values=val)
    assert "kernel=batch-stream" in plan2.describe(), plan2.describe()
    got2 = spmm.csr_spmm(g2["rowptr"], g2["colind"], val, B, plan=plan2)
    assert torch.equal(got2.view(torch.int32), spmm.csr_spmm(g2["rowptr"], g2["colind"], val, B).view(torch.int32))
    assert sampled_rows_equal_oracle(oracle, g2["rowptr"], g2["colind"], val, B, got2, nrows=512, seed=1)
    # N = 32 and 512 through plans of their own
    for N in (32, 512):
        Bn = (torch.randint(0, 100, (M, N), device="cuda", dtype=torch.int32) - 50).float() / 100
        for kernel in ("stream", "seg-stream"):
            pn = spmm.SpmmPlan(rp, ci, M, N, values=val, reorder=True, kernel=kernel)
            assert torch.equal(spmm.csr_spmm(rp, ci, val, Bn, plan=pn).view(torch.int32),
                               spmm.csr_spmm(rp, ci, val, Bn).view(torch.int32)), (N, kernel)


def test_hub_rows_through_a_clustered_plan(pkg):
    """RMAT scale 18 (hub rows of 10^4+ entries): the plan sees the longest row and switches the long-row pass on;
    a plain call with the same decision gives the same bits (chunk sums do not depend on the processing order)."""
    from gespmm_amd import _lib, graphs, spmm

    g = graphs.rmat_shard(18, 16, 0, 1, seed=42, device="cuda")
    rp, ci, M, K = g["rowptr"], g["colind"], g["M"], g["K"]
    assert int((rp[1:] - rp[:-1]).max()) > 4096
    val = torch.rand(g["nnz"], device="cuda") - 0.5
    B = torch.rand(K, 64, device="cuda") - 0.5
    plan = spmm.SpmmPlan(rp, ci, K, 64, values=val, reorder=True)
    assert "long_rows>" in plan.describe(), plan.describe()
    got = spmm.csr_spmm(rp, ci, val, B, plan=plan)
    ref = spmm.csr_spmm(rp, ci, val, B, cfg={"flags": _lib.FLAG_SPLIT_LONG_ROWS})
    assert torch.equal(got.view(torch.int32), ref.view(torch.int32))
    strict = spmm.SpmmPlan(rp, ci, K, 64, values=val, reorder=True, flags=_lib.FLAG_STRICT_ORDER)
    ref_s = spmm.csr_spmm(rp, ci, val, B, cfg={"flags": _lib.FLAG_STRICT_ORDER})
    assert torch.equal(spmm.csr_spmm(rp, ci, val, B, plan=strict).view(torch.int32), ref_s.view(torch.int32))


def test_sddmm_through_a_plan_keeps_the_bits(pkg, bundled):
    """gespmm_plan_sddmm_f32: edges walked in the plan's clustered order, results scattered back to the caller's CSR edge
    order — the same lane butterfly per edge, so the same bits as csr_sddmm / coo_sddmm; also through SPMMFunction's
    edge-weight gradient with plans."""
    import gespmm_amd
    from gespmm_amd import graphs, sddmm, spmm

    cases = [("pubmed", bundled["pubmed"])]
    gs = graphs.synthetic_graph("com-amazon-sbm", seed=42, device="cuda")
    for name, G in cases + [("com-amazon-sbm", gs)]:
        rp = G["rowptr"] if torch.is_tensor(G["rowptr"]) else _dev(G["rowptr"])
        ci = G["colind"] if torch.is_tensor(G["colind"]) else _dev(G["colind"])
        M, K = G["M"], G["K"]
        for N in (3, 41, 128):
            D1 = torch.rand(M, N, device="cuda") - 0.5
            D2 = torch.rand(K, N, device="cuda") - 0.5
            plan = spmm.SpmmPlan(rp, ci, K, N, reorder=True)
            ref = sddmm.csr_sddmm(rp, ci, D1, D2)
            got = sddmm.csr_sddmm(rp, ci, D1, D2, plan=plan)
            assert torch.equal(got.view(torch.int32), ref.view(torch.int32)), (name, N)
            keep = spmm.SpmmPlan(rp, ci, K, N, reorder=False)
            assert torch.equal(sddmm.csr_sddmm(rp, ci, D1, D2, plan=keep).view(torch.int32), ref.view(torch.int32))
    # autograd: edge-weight gradient with and without plans
    G = bundled["pubmed"]
    rp, ci = _dev(G["rowptr"]), _dev(G["colind"])
    colptr, rowind = graphs.transpose_csr(rp, ci)
    w = torch.rand(G["nnz"], device="cuda") - 0.5
    _, _, w_csc = graphs.transpose_csr(rp, ci, val=w)
    x0 = torch.rand(G["M"], 64, device="cuda") - 0.5
    go = torch.rand(G["M"], 64, device="cuda") - 0.5
    grads = []
    for plans in (None, (spmm.SpmmPlan(rp, ci, G["K"], 64, reorder=True), spmm.SpmmPlan(colptr, rowind, G["M"], 64, reorder=True))):
        x = x0.clone().requires_grad_(True)
        ww = w.clone().requires_grad_(True)
        y = gespmm_amd.SPMMFunction.apply(rp, ci, colptr, rowind, x, ww, w_csc, True, plans)
        y.backward(go)
        grads.append((y.detach(), x.grad, ww.grad))
    for a, b in zip(*grads):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))


def _misaligned(t):
    """Same values, storage shifted by one float: 4-byte aligned only."""
    buf = torch.empty(t.numel() + 1, dtype=t.dtype, device=t.device)
    v = buf[1:].view(t.shape)
    v.copy_(t)
    assert v.data_ptr() % 16 != 0
    return v


@pytest.mark.parametrize("kernel", ("auto", "seg-stream", "stream"))
def test_plans_accept_operands_that_are_only_4_byte_aligned(pkg, oracle, bundled, kernel):
    """include/gespmm.h: any N and any 4-byte-aligned B / C are legal. A plan that prefers the segmented-stream kernel
    used to fail (no instantiation for two strips of < 4 floats): it now runs its wavefront task table instead."""
    from gespmm_amd import spmm

    g = bundled["cora"]
    rp, ci = _dev(g["rowptr"]), _dev(g["colind"])
    val_h = oracle.hash_val(g["nnz"], seed=2)
    val = _dev(val_h)
    for N in (128, 200, 66, 12):
        B_h = oracle.hash_B(g["K"], N, seed=N)
        ref = oracle.spmm(g["rowptr"], g["colind"], val_h, B_h, "fma")
        plan = spmm.SpmmPlan(rp, ci, g["K"], N, values=val, reorder=True, kernel=kernel)
        for B in (_dev(B_h), _misaligned(_dev(B_h))):
            for out in (None, _misaligned(torch.empty(g["M"], N, device="cuda"))):
                got = spmm.csr_spmm(rp, ci, val, B, plan=plan, out=out)
                assert np.array_equal(bits(got.cpu().numpy()), bits(ref)), (kernel, N, B.data_ptr() % 16)


def test_kept_split_points_are_not_reused_across_operand_alignments(pkg, oracle):
    """Dense graph through a plan (cache-blocked path, split points kept in the plan's workspace): the slab geometry
    depends on the vector width the operands' alignment allows, so a call with other alignment must rescan."""
    from gespmm_amd import graphs, spmm

    M, deg, N = 20_000, 200, 384
    rp, ci = graphs.synthetic_csr(M, M * deg, symmetric=True, gamma=1.2, seed=9, device="cuda")
    val = torch.rand(ci.numel(), device="cuda") - 0.5
    B = (torch.randint(0, 100, (M, N), device="cuda", dtype=torch.int32) - 50).float() / 100
    plan = spmm.SpmmPlan(rp, ci, M, N, values=val)
    assert "slab" in plan.describe() or "blocked" in plan.describe(), plan.describe()
    ref = spmm.csr_spmm(rp, ci, val, B, cfg={"flags": 0x800 | 0x100})  # streaming, strict order: the CSR-order chain
    a = spmm.csr_spmm(rp, ci, val, B, plan=plan)            # aligned: scans and keeps the split points
    b = spmm.csr_spmm(rp, ci, val, _misaligned(B), plan=plan)  # 4-byte aligned: another slab geometry
    c = spmm.csr_spmm(rp, ci, val, B, plan=plan)            # aligned again
    for got in (a, b, c):
        assert torch.equal(got.view(torch.int32), ref.view(torch.int32))


def test_tune_measures_the_candidates_and_keeps_the_bits(pkg, oracle, bundled):
    """gespmm_plan_tune: the candidate kernels of a clustered plan are timed on the caller's operands and the fastest is kept;
    whichever wins, the product has the oracle's bits; an explicit kernel choice is left alone; another width is refused."""
    from gespmm_amd import _lib, spmm

    g = bundled["pubmed"]
    rp, ci = _dev(g["rowptr"]), _dev(g["colind"])
    val_h = oracle.hash_val(g["nnz"], seed=7)
    val = _dev(val_h)
    for N in (128, 96, 512, 32):
        B_h = oracle.hash_B(g["K"], N, seed=N)
        B = _dev(B_h)
        ref = oracle.spmm(g["rowptr"], g["colind"], val_h, B_h, "fma")
        plan = spmm.SpmmPlan(rp, ci, g["K"], N, values=val, reorder=True)
        got = plan.tune(B, reps=2).cpu().numpy()
        d = plan.describe()
        assert "tuned[us: batch-stream=" in d, d
        times = [float(x.split("=")[1]) for x in d.split("tuned[us: ")[1].split("]")[0].split()]
        assert len(times) == 5 and times[0] > 0 and times[1] > 0, d
        assert times[2] > 0, d  # staged-rows is a candidate at every width since round 6 (96: the general kernel; 32: the lane-group form)
        assert (times[3] > 0) == (N == 32), d          # 4 floats per lane is a candidate at N <= 64
        assert (times[4] > 0) == (N == 32), d          # ... and so is the padded-record kernel (spmm_records.hip)
        assert np.array_equal(bits(got), bits(ref)), (N, d)
        again = spmm.csr_spmm(rp, ci, val, B, plan=plan).cpu().numpy()  # launches after the tune: the kept kernel
        assert np.array_equal(bits(again), bits(ref)), (N, d)
        plan.tune(B)  # tuning twice is allowed (new measurement)
        with pytest.raises(_lib.GespmmError):  # another width than the plan's
            _lib.check(_lib.lib.gespmm_plan_tune(plan._handle, B.data_ptr(), B.data_ptr(), N + 4, 1, None), "gespmm_plan_tune")
    # nothing to measure for an explicit kernel or a storage-order plan — but the product is still returned
    B1_h = oracle.hash_B(g["K"], 128, seed=1)
    ref1 = oracle.spmm(g["rowptr"], g["colind"], val_h, B1_h, "fma")
    explicit = spmm.SpmmPlan(rp, ci, g["K"], 128, values=val, reorder=True, kernel="seg-stream")
    assert np.array_equal(bits(explicit.tune(_dev(B1_h)).cpu().numpy()), bits(ref1))
    assert "tuned[" not in explicit.describe()
    storage = spmm.SpmmPlan(rp, ci, g["K"], 128, values=val, reorder=False)
    assert np.array_equal(bits(storage.tune(_dev(B1_h)).cpu().numpy()), bits(ref1))
    assert "tuned[" not in storage.describe()


def test_cached_memory_limit_and_release(pkg):
    from gespmm_amd import _lib, graphs, spmm

    g = graphs.synthetic_graph("com-amazon-sbm", seed=42, device="cuda", scale=0.25)
    torch.cuda.synchronize()
    for limit in (0, 1 << 30, -1):  # 0: nothing is kept between plans
        _lib.set_cached_memory_limit(limit)
        for _ in range(2):
            plan = spmm.SpmmPlan(g["rowptr"], g["colind"], g["K"], 128, expected_launches=100000)  # (a quarter-size graph: 200 would not pay)
            assert plan.clustered
            del plan
        _lib.release_cached_memory()


def test_plan_options_by_size(pkg, bundled):
    """gespmm_plan_create is the un-versioned symbol: it reads the seven fields every header that shipped with it alone had (reorder ..
    analysis) and nothing beyond them; gespmm_plan_create_v2 reads what the caller's sizeof says, rejects sizes that are not a multiple of
    4, takes defaults for fields the caller does not have and ignores bytes it does not know."""
    import ctypes

    from gespmm_amd import _lib

    lib = _lib.lib
    g = bundled["pubmed"]
    rp, ci = _dev(g["rowptr"]), _dev(g["colind"])
    M, K, nnz = g["M"], g["K"], g["nnz"]

    class Big(ctypes.Structure):  # a FUTURE header: the known fields + two more
        _fields_ = [(n, ctypes.c_int32) for n in ("reorder", "task_entries", "row_floor", "threads", "flags", "kernel", "analysis",
                                                  "expected_launches", "x1", "x2")]

    def create(fn, opt, *size):
        h = ctypes.c_void_p()
        rc = fn(ctypes.byref(h), rp.data_ptr(), ci.data_ptr(), None, M, K, nnz, 128, -1, ctypes.cast(ctypes.byref(opt), ctypes.POINTER(_lib.PlanOptions)),
                *size, None)
        if rc == 0:
            buf = ctypes.create_string_buffer(1024)
            lib.gespmm_plan_describe(h, buf, 1024)
            lib.gespmm_plan_destroy(h)
            return rc, buf.value.decode()
        return rc, ""

    garbage = Big(_lib.PLAN_REORDER, 0, 0, 0, 0, _lib.PLAN_KERNEL_STREAM, 12345, 0, -7, 99)
    rc, d = create(lib.gespmm_plan_create, garbage)  # un-versioned symbol: `analysis` = 12345 IS read (seven fields) -> invalid; x1 / x2 are not
    assert rc == -1, (rc, d)
    host7 = Big(_lib.PLAN_REORDER, 0, 0, 0, 0, _lib.PLAN_KERNEL_STREAM, _lib.PLAN_ANALYSIS_HOST, -5, -7, 99)  # (-5: beyond what this symbol reads)
    rc, d = create(lib.gespmm_plan_create, host7)  # a round-3 caller asking for the host analysis gets it
    assert rc == 0 and "order=clustered" in d and "on the host" in d, (rc, d)
    rc, _ = create(lib.gespmm_plan_create_v2, garbage, ctypes.sizeof(_lib.PlanOptions))  # 28 bytes: `analysis` IS read -> invalid
    assert rc == -1, rc  # GESPMM_EINVAL
    ok = Big(_lib.PLAN_REORDER, 0, 0, 0, 0, _lib.PLAN_KERNEL_STREAM, _lib.PLAN_ANALYSIS_HOST, 300, -7, 99)
    rc, d = create(lib.gespmm_plan_create_v2, ok, ctypes.sizeof(Big))  # a bigger struct than this library knows: the tail is ignored
    assert rc == 0 and "on the host" in d, (rc, d)
    rc, d = create(lib.gespmm_plan_create_v2, ok, 24)  # an old caller through the new symbol: analysis takes its default
    assert rc == 0 and "on the device" in d, (rc, d)
    rc, _ = create(lib.gespmm_plan_create_v2, ok, 26)
    assert rc != 0
    neg = Big(_lib.PLAN_REORDER_AUTO, 0, 0, 0, 0, 0, 0, -3, 0, 0)  # expected_launches < 0 through the symbol that reads it: invalid
    rc, _ = create(lib.gespmm_plan_create_v2, neg, ctypes.sizeof(_lib.PlanOptions))
    assert rc == -1, rc
    removed = Big(_lib.PLAN_REORDER, 0, 0, 0, 0, 2, 0, 0, 0, 0)  # GESPMM_PLAN_KERNEL_LDS_ROWS of rounds 2-3: gone
    rc, _ = create(lib.gespmm_plan_create, removed)
    assert rc != 0
