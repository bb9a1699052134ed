"""Opt-in plan reuse behind the stateless entry points (csrc/auto_plan.cpp, round 6) and the library's warm-up.

The reference's callers pass the CSR arrays with every product and keep nothing (spmmWrapper spmm_test.cu:456-492, spmm_cuda
spmm_kernel.cu:425-458, DGL's CustomCsrmm binary_reduce_sum.cu:338-360). With gespmm_set_auto_plan(k) the library keeps the plan for
them: same bits as the plain call, a pattern or values changed IN PLACE under the same pointers are noticed (a fingerprint of the
arrays before every planned launch), the default (off) is untouched."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _plain(_lib, rp, ci, val, B):
    """The reference product: the _cfg entry point is never planned."""
    M, (K, N) = rp.numel() - 1, B.shape
    C = torch.empty((M, N), device="cuda")
    cfg = _lib.LaunchCfg(0, 0, 0, 0, 0, 0)
    _lib.check(_lib.lib.gespmm_csr_spmm_f32_cfg(_p(rp), _p(ci), _p(val), _p(B), _p(C), M, K, N, ci.numel(), -1, ctypes.byref(cfg), _stream()), "cfg")
    return C


def _call(_lib, rp, ci, val, B):
    M, (K, N) = rp.numel() - 1, B.shape
    C = torch.full((M, N), float("nan"), device="cuda")
    _lib.check(_lib.lib.gespmm_csr_spmm_f32(_p(rp), _p(ci), _p(val), _p(B), _p(C), M, K, N, ci.numel(), -1, _stream()), "gespmm_csr_spmm_f32")
    return C


@pytest.fixture()
def auto(pkg):
    from gespmm_amd import _lib

    _lib.set_auto_plan(0)
    yield _lib
    _lib.set_auto_plan(0)


def test_off_by_default_and_switch_validation(auto, oracle):
    _lib = auto
    assert _lib.auto_plan_stats()["cached_plans"] == 0
    with pytest.raises(_lib.GespmmError):
        _lib.set_auto_plan(-1)


def test_planned_from_the_kth_call_same_bits_and_changes_in_place_are_noticed(auto, oracle):
    _lib = auto
    from gespmm_amd import graphs

    g = graphs.synthetic_graph("com-amazon-sbm", seed=42, device="cuda")
    rp, ci, K, nnz = g["rowptr"], g["colind"], g["K"], g["nnz"]
    val = torch.from_numpy(oracle.hash_val(nnz, seed=3)).cuda()
    B = torch.from_numpy(oracle.hash_B(K, 128, seed=4)).cuda()
    ref = _plain(_lib, rp, ci, val, B)
    before = _lib.auto_plan_stats()
    _lib.set_auto_plan(3)
    outs = [_call(_lib, rp, ci, val, B) for _ in range(5)]
    for o in outs:
        assert torch.equal(o.view(torch.int32), ref.view(torch.int32))
    st = _lib.auto_plan_stats()
    assert st["plans_created"] - before["plans_created"] == 1 and st["cached_plans"] == 1, st
    assert st["calls_planned"] - before["calls_planned"] == 3, st  # calls 3, 4, 5
    # call 3 made the plan behind a synchronous fingerprint; this graph's launches are single kernels, so calls 4 and 5 did not
    # synchronise at all: fingerprint compared on the device, the plan's kernel and the plain kernel behind its verdict
    assert st["fingerprints"] - before["fingerprints"] == 1 and st["calls_async"] - before["calls_async"] == 2, st
    # another dense operand, same key: still the cached plan
    B2 = torch.from_numpy(oracle.hash_B(K, 128, seed=5)).cuda()
    assert torch.equal(_call(_lib, rp, ci, val, B2).view(torch.int32), _plain(_lib, rp, ci, val, B2).view(torch.int32))
    # VALUES changed in place (same pointer): re-permuted before the launch
    val.mul_(-0.5)
    want_v = _plain(_lib, rp, ci, val, B)
    for _ in range(3):  # the call that meets the change runs the plain kernel (device-side verdict); the host re-permutes at a later call
        got = _call(_lib, rp, ci, val, B)
        assert torch.equal(got.view(torch.int32), want_v.view(torch.int32))
        torch.cuda.synchronize()
    assert _lib.auto_plan_stats()["values_refreshed"] - before["values_refreshed"] == 1
    assert torch.equal(_call(_lib, rp, ci, val, B).view(torch.int32), want_v.view(torch.int32))  # (through the refreshed plan)
    # PATTERN changed in place: two entries of different rows swap their columns — same pointers, same nnz, same row lengths
    rph = rp.cpu().numpy()
    r1, r2 = 1000, 200000
    p1, p2 = int(rph[r1]), int(rph[r2])
    assert rph[r1 + 1] > p1 and rph[r2 + 1] > p2
    c1, c2 = int(ci[p1]), int(ci[p2])
    assert c1 != c2
    ci[p1], ci[p2] = c2, c1
    want = _plain(_lib, rp, ci, val, B)
    assert not torch.equal(want.view(torch.int32), want_v.view(torch.int32))  # (the product did change)
    for _ in range(2):  # the first call after the change is served by the plain kernel behind the guard, the host drops the plan at the second
        got = _call(_lib, rp, ci, val, B)
        assert torch.equal(got.view(torch.int32), want.view(torch.int32))
        torch.cuda.synchronize()
    st2 = _lib.auto_plan_stats()
    assert st2["invalidated"] - before["invalidated"] == 1 and st2["cached_plans"] == 0, st2
    # ... and the key earns a new plan after k more calls
    for _ in range(3):
        got = _call(_lib, rp, ci, val, B)
        assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    assert _lib.auto_plan_stats()["cached_plans"] == 1
    # the second plan of this key is not disturbed by what the checks of the first one left behind
    made = _lib.auto_plan_stats()["plans_created"]
    for _ in range(6):
        got = _call(_lib, rp, ci, val, B)
        assert torch.equal(got.view(torch.int32), want.view(torch.int32))
        torch.cuda.synchronize()
    st3 = _lib.auto_plan_stats()
    assert st3["cached_plans"] == 1 and st3["plans_created"] == made and st3["invalidated"] == st2["invalidated"], st3
    _lib.auto_plan_clear()
    assert _lib.auto_plan_stats()["cached_plans"] == 0


def test_dgl_entry_points_and_the_max_reducer(auto, oracle):
    _lib = auto
    from gespmm_amd import graphs, spmm

    g = graphs.synthetic_graph("com-amazon-sbm", seed=42, device="cuda")
    rp, ci, M, K = g["rowptr"], g["colind"], g["M"], g["K"]
    N = 128
    B = torch.from_numpy(oracle.hash_B(K, N, seed=6)).cuda()
    ref_sum = _plain(_lib, rp, ci, None, B)
    ref_max = spmm.csr_spmm_max(rp, ci, B)
    before = _lib.auto_plan_stats()
    _lib.set_auto_plan(2)
    for fn, ref in ((_lib.lib.gespmm_dgl_csrmm_sum_f32, ref_sum), (_lib.lib.gespmm_dgl_csrmm_max_f32, ref_max)):
        for _ in range(4):
            C = torch.full((M, N), float("nan"), device="cuda")
            _lib.check(fn(M, N, _p(rp), _p(ci), _p(B), _p(C), _stream()), "dgl")
            assert torch.equal(C.view(torch.int32), ref.view(torch.int32))
    st = _lib.auto_plan_stats()
    assert st["plans_created"] - before["plans_created"] == 2 and st["calls_planned"] - before["calls_planned"] == 6, st


def test_synchronous_mode_where_a_launch_is_several_kernels(auto, oracle):
    """Hub rows make the plan's launch two kernels and the long-row pass: such a key keeps the synchronous fingerprint (one per planned
    call), and changes in place are noticed by the very call that meets them."""
    _lib = auto
    rng = np.random.RandomState(5)
    M = K = 60000
    degs = rng.randint(2, 24, size=M)
    degs[123] = 5000
    degs[40000] = 3000
    rowptr = np.zeros(M + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(degs)
    rows = np.repeat(np.arange(M), degs)
    colind = ((rows // 64) * 64 + rng.randint(0, 64, size=rows.size)).astype(np.int32) % K  # blocks of 64 rows share their columns
    rp, ci = torch.from_numpy(rowptr).cuda(), torch.from_numpy(colind).cuda()
    val = torch.from_numpy(oracle.hash_val(colind.size, seed=2)).cuda()
    B = torch.from_numpy(oracle.hash_B(K, 128, seed=3)).cuda()
    before = _lib.auto_plan_stats()
    _lib.set_auto_plan(1)
    ref = _plain(_lib, rp, ci, val, B)
    outs = [_call(_lib, rp, ci, val, B) for _ in range(3)]
    for o in outs:
        assert torch.equal(o.view(torch.int32), ref.view(torch.int32))
    st = _lib.auto_plan_stats()
    if st["cached_plans"] == 1:  # (the analysis may decline a matrix this small; the bits hold either way)
        assert st["calls_async"] == before["calls_async"] and st["fingerprints"] - before["fingerprints"] == 3, st
        val.mul_(2.0)
        assert torch.equal(_call(_lib, rp, ci, val, B).view(torch.int32), _plain(_lib, rp, ci, val, B).view(torch.int32))
        assert _lib.auto_plan_stats()["values_refreshed"] - before["values_refreshed"] == 1


def test_no_structure_no_plan_no_fingerprint(auto, oracle):
    """A matrix whose analysis keeps the storage order (here: the cost rule declines) is remembered as such: the plain path, without a
    fingerprint or a synchronisation, from then on."""
    _lib = auto
    from gespmm_amd import graphs

    g = graphs.synthetic_graph("com-amazon-like", seed=42, device="cuda")
    rp, ci, K = g["rowptr"], g["colind"], g["K"]
    B = torch.from_numpy(oracle.hash_B(K, 128, seed=7)).cuda()
    ref = _plain(_lib, rp, ci, None, B)
    _lib.set_auto_plan(1)
    before = _lib.auto_plan_stats()
    for _ in range(4):
        assert torch.equal(_call(_lib, rp, ci, None, B).view(torch.int32), ref.view(torch.int32))
    st = _lib.auto_plan_stats()
    assert st["cached_plans"] == 0 and st["calls_planned"] == before["calls_planned"], st
    assert st["fingerprints"] - before["fingerprints"] == 1, st  # only the call that asked the analysis


def test_init_is_idempotent_and_warms_the_analysis(pkg):
    from gespmm_amd import _lib

    _lib.init()
    _lib.init(334863, 1851744)
    # after the warm-up the cost rule no longer adds the cold cost (the query's cold_start is what a cold process would send)
    warm = _lib.plan_policy(334863, 334863, 1851744, 128, 100, wedge_probe=0.42)
    cold = _lib.plan_policy(334863, 334863, 1851744, 128, 100, wedge_probe=0.42, cold_start=1)
    assert cold["est_cost_us"] - warm["est_cost_us"] == pytest.approx(29000.0)
    assert warm["analyse"] == 1 and cold["analyse"] == 0


def test_narrow_width_through_the_record_kernel(auto, oracle):
    """N = 32 on the headline graph: a cached plan launches the padded-record kernel (spmm_records.hip) behind the device-side guard; same
    bits as the plain call, values changed in place are noticed and the records refilled."""
    _lib = auto
    from gespmm_amd import graphs

    g = graphs.synthetic_graph("com-amazon-sbm", seed=42, device="cuda")
    rp, ci, K, nnz = g["rowptr"], g["colind"], g["K"], g["nnz"]
    val = torch.from_numpy(oracle.hash_val(nnz, seed=3)).cuda()
    B = torch.from_numpy(oracle.hash_B(K, 32, seed=4)).cuda()
    ref = _plain(_lib, rp, ci, val, B)
    before = _lib.auto_plan_stats()
    _lib.set_auto_plan(2)
    for _ in range(5):
        assert torch.equal(_call(_lib, rp, ci, val, B).view(torch.int32), ref.view(torch.int32))
    val.mul_(-0.25)
    want = _plain(_lib, rp, ci, val, B)
    for _ in range(4):
        assert torch.equal(_call(_lib, rp, ci, val, B).view(torch.int32), want.view(torch.int32))
        torch.cuda.synchronize()
    st = _lib.auto_plan_stats()
    if st["cached_plans"] == 1:  # (the cost rule may keep the storage order inside 200 launches at this width: then nothing was planned)
        assert st["calls_planned"] - before["calls_planned"] >= 3, st
        assert st["values_refreshed"] - before["values_refreshed"] == 1, st
    _lib.auto_plan_clear()
