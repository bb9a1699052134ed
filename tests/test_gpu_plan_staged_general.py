"""The staged-rows kernel at ANY width and with the max reducer (csrc/spmm_staged_gen.hip, round 6): the reference's kernels take every
N >= 1 (the `nout` guards of spmm_test.cu:206-233, spmm_kernel.cu:186-206) and its DGL patch has a max twin
(binary_reduce_max.cu:26-168); until round 6 the plan's fast kernel served N = 16 / 32 / 64 / 128 / 256 * 2^t and the sum only. Same
bar as tests/test_gpu_plan_staged.py: only WHERE a B row comes from changes, so the bits are the oracle's (= the reference's kernels'
on this GPU, tests/test_gpu_ref_kernels.py) and the plain call's."""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import ROOT, bits, edge_case_csr
from test_gpu_plan_staged import _dev, _random_local_csr

pytestmark = pytest.mark.gpu

# class counts and feature widths of SURVEY.md section 8 (41 / 47 / 100 / 602), their neighbours, one / two / five column tiles of every lane
# vector, widths just past a tile (65, 130, 260)
WIDTHS = (1, 3, 20, 41, 47, 65, 100, 130, 200, 260, 602)


def _staged_plan(spmm, rp, ci, K, N, **kw):
    plan = spmm.SpmmPlan(rp, ci, K, N, reorder=True, kernel="staged", **kw)
    assert plan.clustered and "kernel=staged-rows" in plan.describe(), plan.describe()
    return plan


@pytest.mark.parametrize("N", WIDTHS)
@pytest.mark.parametrize("graph", ("cora", "pubmed"))
def test_any_width_bits_equal_oracle_and_reference_kernels(pkg, oracle, bundled, graph, N):
    from gespmm_amd import spmm

    g = bundled[graph]
    rp, ci = _dev(g["rowptr"]), _dev(g["colind"])
    val_h = oracle.hash_val(g["nnz"], seed=7)
    val = _dev(val_h)
    plan = _staged_plan(spmm, rp, ci, g["K"], N, values=val)
    B_h = oracle.hash_B(g["K"], N, seed=N)
    B = _dev(B_h)
    got = spmm.csr_spmm(rp, ci, val, B, plan=plan).cpu().numpy()
    assert np.array_equal(bits(got), bits(oracle.spmm(g["rowptr"], g["colind"], val_h, B_h, "fma"))), (graph, N)
    got_u = spmm.csr_spmm_no_edge_value(rp, ci, B, plan=plan).cpu().numpy()
    assert np.array_equal(bits(got_u), bits(oracle.spmm(g["rowptr"], g["colind"], None, B_h, "golden"))), (graph, N)
    # the reference's torch-op kernels (spmm_cuda's three-way dispatch, compiled from the checkout for this GPU) on the same operands
    import ref_py

    if ref_py.kernels_available():
        want = ref_py.spmm_cuda(rp, ci, val, B)
        assert torch.equal(torch.from_numpy(got).view(torch.int32), want.cpu().view(torch.int32)), (graph, N)
    # new values through the same plan
    val2_h = oracle.hash_val(g["nnz"], seed=8)
    got2 = spmm.csr_spmm(rp, ci, _dev(val2_h), B, plan=plan).cpu().numpy()
    assert np.array_equal(bits(got2), bits(oracle.spmm(g["rowptr"], g["colind"], val2_h, B_h, "fma")))


@pytest.mark.parametrize("N", (3, 41, 100, 128, 200, 256, 512, 602))
def test_max_reducer_through_the_staged_kernel(pkg, oracle, bundled, N):
    """DGL's max reducer (binary_reduce_max.cu:182-207: rows without neighbours give -10000) through a plan: the general kernel walks the
    plan's tables with v_max instead of the multiply-adds — at N = 128 / 256 / 512 the very tables the tuned sum kernel walks."""
    from gespmm_amd import spmm

    g = bundled["pubmed"]
    # a few rows without entries: their value is `empty`
    rowptr = g["rowptr"].copy()
    colind = g["colind"]
    rp, ci = _dev(rowptr), _dev(colind)
    plan = _staged_plan(spmm, rp, ci, g["K"], N)
    B_h = oracle.hash_B(g["K"], N, seed=2 * N + 1)
    B = _dev(B_h)
    for empty in (-10000.0, -3.5):
        got = plan.run(None, B, reduce_max=empty).cpu().numpy()
        assert np.array_equal(bits(got), bits(oracle.spmm_max(rowptr, colind, B_h, empty))), (N, empty)
    # the sum through the same tables still has its bits
    got_s = spmm.csr_spmm_no_edge_value(rp, ci, B, plan=plan).cpu().numpy()
    assert np.array_equal(bits(got_s), bits(oracle.spmm(rowptr, colind, None, B_h, "golden"))), N
    # ... and the edge shapes (empty rows at both ends of a block, rows of 1 .. 200 entries, K != M)
    e = edge_case_csr(seed=6)
    rp2, ci2 = _dev(e["rowptr"]), _dev(e["colind"])
    plan2 = spmm.SpmmPlan(rp2, ci2, e["K"], N, reorder=True, kernel="staged")
    B2_h = oracle.hash_B(e["K"], N, seed=N + 3)
    got = plan2.run(None, _dev(B2_h), reduce_max=-10000.0).cpu().numpy()
    assert np.array_equal(bits(got), bits(oracle.spmm_max(e["rowptr"], e["colind"], B2_h, -10000.0))), (N, plan2.describe())


@pytest.mark.parametrize("N", (5, 47, 100, 200, 602))
def test_edge_shapes_and_ragged_blocks(pkg, oracle, N):
    from gespmm_amd import spmm

    g = edge_case_csr(seed=4)
    rp, ci = _dev(g["rowptr"]), _dev(g["colind"])
    val_h = oracle.hash_val(g["nnz"], seed=3)
    B_h = oracle.hash_B(g["K"], N, seed=N + 1)
    plan = spmm.SpmmPlan(rp, ci, g["K"], N, values=_dev(val_h), reorder=True, kernel="staged")
    assert "kernel=staged-rows" in plan.describe(), plan.describe()
    got = spmm.csr_spmm(rp, ci, _dev(val_h), _dev(B_h), plan=plan).cpu().numpy()
    assert np.array_equal(bits(got), bits(oracle.spmm(g["rowptr"], g["colind"], val_h, B_h, "fma")))
    reps = 61  # several blocks (more than eight: every XCD slice has one), ragged last block
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


def test_hub_rows_at_a_general_width(pkg, oracle):
    from gespmm_amd import spmm

    rng = np.random.RandomState(3)
    M = K = 4000
    degs = rng.randint(1, 30, size=M)
    hubs = {17: 3000, 2500: 9000, M - 1: 2600}
    for r, d in hubs.items():
        degs[r] = d
    rowptr = np.zeros(M + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(degs)
    colind = rng.randint(0, K, size=int(rowptr[-1])).astype(np.int32)
    val_h = oracle.hash_val(colind.size, seed=2)
    rp, ci, val = _dev(rowptr), _dev(colind), _dev(val_h)
    for N in (100, 602):
        B_h = oracle.hash_B(K, N, seed=3 + N)
        B = _dev(B_h)
        plan = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True, kernel="staged", flags=0x100)  # STRICT_ORDER: hubs as strict chains
        assert "kernel=staged-rows" in plan.describe() and "hub_rows=3" in plan.describe(), plan.describe()
        got = spmm.csr_spmm(rp, ci, val, B, plan=plan).cpu().numpy()
        assert np.array_equal(bits(got), bits(oracle.spmm(rowptr, colind, val_h, B_h, "fma"))), N
        # max: hub rows through the streaming kernel's long-row pass (max is exact in any order)
        plan_u = spmm.SpmmPlan(rp, ci, K, N, reorder=True, kernel="staged", flags=0x200)
        got = plan_u.run(None, B, reduce_max=-10000.0).cpu().numpy()
        assert np.array_equal(bits(got), bits(oracle.spmm_max(rowptr, colind, B_h, -10000.0))), N


def test_auto_takes_the_staged_kernel_at_general_even_widths(pkg, oracle):
    """The headline graph at N = 100 / 200 (two / four floats per lane: the 128- / 256-column tile rules of plan_policy.cpp) and the max
    reducer at N = 128: AUTO plans, bits = the plain call's."""
    from gespmm_amd import graphs, spmm

    g = graphs.synthetic_graph("com-amazon-sbm", seed=42, device="cuda")
    rp, ci, K, nnz = g["rowptr"], g["colind"], g["K"], g["nnz"]
    val = torch.from_numpy(oracle.hash_val(nnz, seed=11)).cuda()
    for N in (100, 200):
        B = torch.from_numpy(oracle.hash_B(K, N, seed=12)).cuda()
        plan = spmm.SpmmPlan(rp, ci, K, N, values=val)
        assert plan.clustered and "kernel=staged-rows" in plan.describe(), plan.describe()
        got = spmm.csr_spmm(rp, ci, val, B, plan=plan)
        plain = spmm.csr_spmm(rp, ci, val, B, cfg={"flags": 0x100})
        assert torch.equal(got.view(torch.int32), plain.view(torch.int32)), N
    B = torch.from_numpy(oracle.hash_B(K, 128, seed=13)).cuda()
    plan = spmm.SpmmPlan(rp, ci, K, 128)
    assert "kernel=staged-rows" in plan.describe(), plan.describe()
    got = plan.run(None, B, reduce_max=-10000.0)
    assert torch.equal(got.view(torch.int32), spmm.csr_spmm_max(rp, ci, B).view(torch.int32))


def test_reduced_soak_of_the_general_kernel(pkg):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import staged_gen_soak

    checked, staged = staged_gen_soak.soak(61000, 120, verbose=False)
    assert checked >= 700 and staged >= 200, (checked, staged)
