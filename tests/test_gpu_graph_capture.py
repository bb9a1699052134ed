"""HIP graph capture of the product launches (torch.cuda.CUDAGraph = hipGraph on ROCm): the launch paths allocate nothing and never
synchronise, so a caller with a launch-bound inner loop (the reference's GCN epoch: gcn_custom.py:120-143, two propagations per layer pass)
can capture them once and replay. Replays must see NEW operand contents (same addresses) and give the eager call's bits."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _capture(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):  # warm-up on the side stream (code objects, the plan's scratch)
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = fn()
    return graph, out


@pytest.mark.parametrize("N,kernel,tag", ((128, "staged", "kernel=staged-rows"), (32, "records", "kernel=padded-records"), (47, "records", "kernel=padded-records"),
                                           (64, "stream", "batch-stream"), (96, "seg-stream", "segmented-stream")))
def test_plan_launches_replay_from_a_graph(pkg, oracle, bundled, N, kernel, tag):
    from gespmm_amd import spmm

    g = bundled["pubmed"]
    rp, ci = _dev(g["rowptr"]), _dev(g["colind"])
    val_h = oracle.hash_val(g["nnz"], seed=7)
    val = _dev(val_h)
    plan = spmm.SpmmPlan(rp, ci, g["K"], N, values=val, reorder=True, kernel=kernel)
    assert tag in plan.describe(), plan.describe()
    B = _dev(oracle.hash_B(g["K"], N, seed=1))
    C = torch.empty((g["M"], N), device="cuda")
    graph, _ = _capture(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan))
    for seed in (2, 3):
        B_h = oracle.hash_B(g["K"], N, seed=seed)
        B.copy_(_dev(B_h))  # new contents, same address
        C.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(C.cpu().numpy().view(np.uint32), oracle.spmm(g["rowptr"], g["colind"], val_h, B_h, "fma").view(np.uint32)), (N, kernel, seed)


def test_plain_calls_and_a_two_layer_propagation_replay_from_a_graph(pkg, oracle, bundled):
    """The stateless entry points (no plan) under capture — incl. a launch that takes the long-row pass with the workspace the Python layer
    hands it (a torch allocation: under capture a block of the graph's private pool; without a workspace the library switches the pass off
    while capturing) — and the shape of a GCN forward: two products of different widths in one graph. (Until the end of round 6 this test
    failed in about every second FRESH process: the pass zeroed its header with a captured memset — DESIGN 3.8b, profiles/r06/capture_flake/.)"""
    from gespmm_amd import _lib, spmm

    g = bundled["cora"]
    rp, ci = _dev(g["rowptr"]), _dev(g["colind"])
    val_h = oracle.hash_val(g["nnz"], seed=5)
    val = _dev(val_h)
    X = _dev(oracle.hash_B(g["K"], 128, seed=1))
    H = torch.empty((g["M"], 128), device="cuda")
    Y = torch.empty((g["M"], 7), device="cuda")
    plan_h = spmm.SpmmPlan(rp, ci, g["K"], 128, values=val)
    cfg = {"flags": _lib.FLAG_SPLIT_LONG_ROWS}

    def forward():
        spmm.csr_spmm(rp, ci, val, X, out=H, plan=plan_h)            # layer 1 through a plan
        spmm.csr_spmm(rp, ci, val, H[:, :7].contiguous(), out=Y, cfg=cfg)  # layer 2: plain call, long-row pass forced (pool scratch)
        return Y

    graph, _ = _capture(forward)
    for seed in (2, 3):
        X_h = oracle.hash_B(g["K"], 128, seed=seed)
        X.copy_(_dev(X_h))
        H.zero_()
        Y.zero_()
        graph.replay()
        torch.cuda.synchronize()
        H_ref = oracle.spmm(g["rowptr"], g["colind"], val_h, X_h, "fma")
        assert np.array_equal(H.cpu().numpy().view(np.uint32), H_ref.view(np.uint32))
        Y_ref = oracle.spmm(g["rowptr"], g["colind"], val_h, np.ascontiguousarray(H_ref[:, :7]), "fma")
        got = Y.cpu().numpy()
        # (the long-row pass re-associates rows beyond its threshold: cora has none, so the bits are the oracle's)
        assert np.array_equal(got.view(np.uint32), Y_ref.view(np.uint32))
