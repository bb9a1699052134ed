"""The plan's padded-record kernel (csrc/spmm_records.hip): narrow widths (4 <= N <= 64), rows of <= 1024 entries, sum reducer.
A row is cut into pieces of 8 padded entry slots that stay in ONE lane group's chain, in order — every output element is still one
fp32 accumulator over the row's entries in ascending CSR position with one fused multiply-add per entry (the reference's kernels:
spmm_test.cu:182-203) — so the bits must equal the oracle's `fma` arithmetic and the plain call's, whatever the padding."""
import numpy as np
import pytest
import torch

from helpers import bits, edge_case_csr

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("N", (4, 5, 7, 8, 12, 16, 17, 20, 30, 32, 36, 41, 47, 48, 63, 64))  # lane groups of 4 (N <= 16), 8 (N <= 32), 16 (N <= 64); widths that are
# not multiples of 4: 4-byte-aligned vectors, the lane at the row's end takes the last four columns
@pytest.mark.parametrize("reorder", (True, False))
@pytest.mark.parametrize("graph", ("cora", "pubmed"))
def test_bits_equal_oracle_valued_unweighted_and_new_values(pkg, oracle, bundled, graph, reorder, N):
    from gespmm_amd import spmm

    g = bundled[graph]
    rp, ci = _dev(g["rowptr"]), _dev(g["colind"])
    val_h = oracle.hash_val(g["nnz"], seed=7)
    val = _dev(val_h)
    plan = spmm.SpmmPlan(rp, ci, g["K"], N, values=val, reorder=reorder, kernel="records")
    assert "kernel=padded-records" in plan.describe(), plan.describe()
    assert plan.clustered == reorder
    B_h = oracle.hash_B(g["K"], N, seed=N)
    B = _dev(B_h)
    got = spmm.csr_spmm(rp, ci, val, B, plan=plan).cpu().numpy()
    assert np.array_equal(bits(got), bits(oracle.spmm(g["rowptr"], g["colind"], val_h, B_h, "fma"))), (graph, N)
    # the same plan without values: the records carry 1.0f (fma(1, b, acc) == acc + b), against the golden loop
    got_u = spmm.csr_spmm_no_edge_value(rp, ci, B, plan=plan).cpu().numpy()
    assert np.array_equal(bits(got_u), bits(oracle.spmm(g["rowptr"], g["colind"], None, B_h, "golden"))), (graph, N)
    # with other values (the records are refilled on the device)
    val2_h = oracle.hash_val(g["nnz"], seed=8)
    val2 = _dev(val2_h)
    got2 = spmm.csr_spmm(rp, ci, val2, B, plan=plan).cpu().numpy()
    assert np.array_equal(bits(got2), bits(oracle.spmm(g["rowptr"], g["colind"], val2_h, B_h, "fma")))
    # what the record kernel does not serve goes to the streaming kernels of the same plan: another width, the max reducer
    B2_h = oracle.hash_B(g["K"], 128, seed=5)
    got128 = spmm.csr_spmm(rp, ci, val2, _dev(B2_h), plan=plan).cpu().numpy()
    assert np.array_equal(bits(got128), bits(oracle.spmm(g["rowptr"], g["colind"], val2_h, B2_h, "fma")))
    plan_u = spmm.SpmmPlan(rp, ci, g["K"], N, reorder=reorder, kernel="records")  # (the max reducer takes plans without values)
    got_max = plan_u.run(None, B, reduce_max=-10000.0).cpu().numpy()
    assert np.array_equal(bits(got_max), bits(oracle.spmm_max(g["rowptr"], g["colind"], B_h, -10000.0))), (graph, N)


@pytest.mark.parametrize("N", (16, 32, 47, 64))
def test_edge_shapes(pkg, oracle, N):
    """Empty rows (leading, trailing, runs of them), rows of 1..200 entries (1 to 25 pieces in one chain), repeated and unsorted
    columns, K != M, M smaller than one task and not a multiple of the rows per task."""
    from gespmm_amd import spmm

    g = edge_case_csr(seed=4)
    rp, ci = _dev(g["rowptr"]), _dev(g["colind"])
    val_h = oracle.hash_val(g["nnz"], seed=3)
    B_h = oracle.hash_B(g["K"], N, seed=N + 1)
    plan = spmm.SpmmPlan(rp, ci, g["K"], N, values=_dev(val_h), reorder=True, kernel="records")
    assert "kernel=padded-records" in plan.describe(), plan.describe()
    got = spmm.csr_spmm(rp, ci, _dev(val_h), _dev(B_h), plan=plan).cpu().numpy()
    assert np.array_equal(bits(got), bits(oracle.spmm(g["rowptr"], g["colind"], val_h, B_h, "fma")))
    # many tasks, ragged last task: the same rows tiled 37 times with shifted columns
    reps = 37
    degs = np.tile(np.diff(g["rowptr"]), reps)
    rowptr = np.zeros(degs.size + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(degs)
    colind = np.concatenate([(g["colind"] + 7 * r) % g["K"] for r in range(reps)]).astype(np.int32)
    val_h = oracle.hash_val(colind.size, seed=5)
    rp2, ci2 = _dev(rowptr), _dev(colind)
    for reorder in (True, False):
        plan = spmm.SpmmPlan(rp2, ci2, g["K"], N, values=_dev(val_h), reorder=reorder, kernel="records")
        assert "kernel=padded-records" in plan.describe(), plan.describe()
        got = spmm.csr_spmm(rp2, ci2, _dev(val_h), _dev(B_h), plan=plan).cpu().numpy()
        assert np.array_equal(bits(got), bits(oracle.spmm(rowptr, colind, val_h, B_h, "fma"))), reorder


def test_special_values_do_not_leak_through_the_padding(pkg, oracle):
    """Padded slots repeat the address of the piece's last entry and are predicated off: an inf / nan in B must reach exactly the rows
    that reference it, as in the plain call (a multiply by a padded 0 would turn inf into nan)."""
    from gespmm_amd import spmm

    g = edge_case_csr(seed=9)
    N = 32
    rp, ci = _dev(g["rowptr"]), _dev(g["colind"])
    val_h = oracle.hash_val(g["nnz"], seed=1)
    B_h = oracle.hash_B(g["K"], N, seed=2).copy()
    B_h[0, :] = np.inf       # column 0 is also what an empty slot points at
    B_h[g["colind"][-1], 3] = -np.inf
    B_h[g["colind"][5], 7] = np.nan
    val, B = _dev(val_h), _dev(B_h)
    plan = spmm.SpmmPlan(rp, ci, g["K"], N, values=val, reorder=True, kernel="records")
    assert "kernel=padded-records" in plan.describe(), plan.describe()
    got = spmm.csr_spmm(rp, ci, val, B, plan=plan).cpu().numpy()
    ref = spmm.csr_spmm(rp, ci, val, B).cpu().numpy()
    assert np.array_equal(bits(got), bits(ref))


def test_rows_beyond_the_limit_and_unaligned_operands_fall_back(pkg, oracle):
    from gespmm_amd import spmm

    rng = np.random.RandomState(3)
    M, K, N = 300, 5000, 32
    degs = rng.randint(0, 12, size=M)
    degs[17] = 1500  # beyond kRecordMaxRow = 1024: the tables are not built, the streaming kernels serve the plan
    rowptr = np.zeros(M + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(degs)
    colind = rng.randint(0, K, size=int(rowptr[-1])).astype(np.int32)
    val_h = oracle.hash_val(colind.size, seed=2)
    B_h = oracle.hash_B(K, N, seed=4)
    rp, ci, val = _dev(rowptr), _dev(colind), _dev(val_h)
    plan = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True, kernel="records")
    assert "kernel=padded-records" not in plan.describe(), plan.describe()
    got = spmm.csr_spmm(rp, ci, val, _dev(B_h), plan=plan).cpu().numpy()
    assert np.array_equal(bits(got), bits(oracle.spmm(rowptr, colind, val_h, B_h, "fma")))
    # a B that starts 4 bytes into an allocation: the record kernel wants 16-byte rows, the plan launches a streaming kernel
    degs[17] = 9
    rowptr[1:] = np.cumsum(degs)
    colind = colind[: int(rowptr[-1])]
    val_h = val_h[: colind.size]
    rp, ci, val = _dev(rowptr), _dev(colind), _dev(val_h)
    plan = spmm.SpmmPlan(rp, ci, K, N, values=val, reorder=True, kernel="records")
    assert "kernel=padded-records" in plan.describe(), plan.describe()
    flat = torch.empty(K * N + 1, device="cuda")
    Bu = flat[1:].view(K, N)
    Bu.copy_(_dev(B_h))
    got = spmm.csr_spmm(rp, ci, val, Bu, plan=plan).cpu().numpy()
    assert np.array_equal(bits(got), bits(oracle.spmm(rowptr, colind, val_h, B_h, "fma")))


def test_reduced_soak_of_the_record_kernel(pkg):
    """120 seeds of scripts/records_soak.py (random shapes, widths 4 .. 64, both orders, rows at and beyond the 1024-entry limit,
    valued / unweighted / new values) against the plain call's strict-order bits."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import records_soak

    checked, served = records_soak.soak(71000, 120, verbose=False)
    assert checked >= 1200 and served >= 330, (checked, served)


def test_a_matrix_that_arrives_clustered_keeps_its_order_and_takes_the_record_kernel(pkg, oracle):
    """The headline graph relabelled in its planted order: the plan judges the storage order as good as its own clustering, keeps it
    (no permuted copy) — and AUTO still launches the padded-record kernel at N = 32, by the modelled hits of the storage order."""
    from gespmm_amd import graphs, spmm
    from helpers import sampled_rows_equal_oracle

    g = graphs.synthetic_graph("com-amazon-sbm", seed=42, device="cuda")
    rp, ci = graphs.relabel_by_order(g["rowptr"], g["colind"], torch.argsort(g["truth"]))
    val = torch.from_numpy(oracle.hash_val(g["nnz"], seed=7)).cuda()
    B = torch.from_numpy(oracle.hash_B(g["K"], 32, seed=3)).cuda()
    plan = spmm.SpmmPlan(rp, ci, g["K"], 32, values=val, expected_launches=1000000)
    d = plan.describe()
    got = spmm.csr_spmm(rp, ci, val, B, plan=plan)
    assert torch.equal(got.view(torch.int32), spmm.csr_spmm(rp, ci, val, B).view(torch.int32)), d
    assert sampled_rows_equal_oracle(oracle, rp, ci, val, B, got, nrows=256, seed=1), d
    if not plan.clustered:  # (the model may also prefer its own order by a few points: then the rule is the one the audit covers)
        assert "kernel=padded-records" in d, d


def test_a_matrix_that_arrives_clustered_gets_the_staged_kernel_at_wide_widths(pkg, oracle):
    """Same matrix at N = 128: the plan keeps the caller's order but makes its own copy in it, because the staged-rows kernel walks the
    plan's tables (until round 6 such a matrix kept the streaming kernels: 122 against 82 us)."""
    from gespmm_amd import graphs, spmm
    from helpers import sampled_rows_equal_oracle

    g = graphs.synthetic_graph("com-amazon-sbm", seed=42, device="cuda")
    rp, ci = graphs.relabel_by_order(g["rowptr"], g["colind"], torch.argsort(g["truth"]))
    val = torch.from_numpy(oracle.hash_val(g["nnz"], seed=7)).cuda()
    B = torch.from_numpy(oracle.hash_B(g["K"], 128, seed=3)).cuda()
    plan = spmm.SpmmPlan(rp, ci, g["K"], 128, values=val, expected_launches=1000000)
    d = plan.describe()
    assert "kernel=staged-rows" in d, d
    got = spmm.csr_spmm(rp, ci, val, B, plan=plan)
    assert torch.equal(got.view(torch.int32), spmm.csr_spmm(rp, ci, val, B).view(torch.int32)), d
    assert sampled_rows_equal_oracle(oracle, rp, ci, val, B, got, nrows=256, seed=2), d
    if "order=storage(plan copy" in d:
        assert plan.order().tolist()[:5] == [0, 1, 2, 3, 4]
    # new values through the same plan
    val2 = torch.from_numpy(oracle.hash_val(g["nnz"], seed=9)).cuda()
    got2 = spmm.csr_spmm(rp, ci, val2, B, plan=plan)
    assert torch.equal(got2.view(torch.int32), spmm.csr_spmm(rp, ci, val2, B).view(torch.int32))
