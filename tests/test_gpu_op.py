"""SPMMFunction / GCNConv (mirror of pytorch-custom/op.py) against a dense-torch
restatement on the CPU:  out = D_in^-1/2 · A · (D_out^-1/2 ⊙ (X W)) + b,
grad_feat = A^T · grad_out  (SURVEY.md §8 c2, A4), with A built exactly as
gcn_custom.py:29-49 builds it (self-loops added, CSR stored as colptr/rowind, CSC as
rowptr/colind)."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

pytestmark = pytest.mark.gpu


def proc_like_reference(edge_index, n_v, add_self_loop=True):
    """Host-side restatement of what the reference's caller hands the op (gcn_custom.py:29-49): A = adjacency of
    (src, dst) pairs plus the identity, duplicates summed; `colptr/rowind/value_csc` describe A row-major (by source),
    `rowptr/colind/value_csr` describe A^T row-major (by destination). Returns host arrays and the dense A."""
    src, dst = np.asarray(edge_index[0], np.int64), np.asarray(edge_index[1], np.int64)
    A = sp.csr_matrix((np.ones(src.size), (src, dst)), shape=(n_v, n_v))
    if add_self_loop:
        A = A + sp.identity(n_v, format="csr")
    A.sum_duplicates()
    A.sort_indices()
    At = A.T.tocsr()
    At.sort_indices()
    return {"colptr": A.indptr, "rowind": A.indices, "value_csc": A.data.astype(np.float32),
            "rowptr": At.indptr, "colind": At.indices, "value_csr": At.data.astype(np.float32),
            "dense": A.toarray().astype(np.float64)}


def test_example_proc_builds_the_same_operands(graph):
    """examples/gcn_custom.py builds both index orders on the device with torch ops; same arrays as the scipy route."""
    import importlib.util
    import os

    from helpers import ROOT

    spec = importlib.util.spec_from_file_location("gcn_example", os.path.join(ROOT, "examples", "gcn_custom.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = mod.proc(graph["edge_index"], graph["n_v"], "cuda")
    for k in ("colptr", "rowind", "value_csc", "rowptr", "colind", "value_csr"):
        assert np.array_equal(g[k].cpu().numpy(), graph[k]), k


@pytest.fixture(scope="module")
def graph():
    rng = np.random.RandomState(0)
    n_v, n_e = 300, 2400
    src = rng.randint(0, n_v, n_e)
    dst = rng.randint(0, n_v, n_e)
    keep = src != dst
    ei = np.unique(np.stack([src[keep], dst[keep]]), axis=1).astype(np.int32)
    g = proc_like_reference(ei, n_v)
    for k in ("rowptr", "colind", "colptr", "rowind"):
        g[k + "_d"] = torch.from_numpy(g[k].astype(np.int32)).cuda()
    g["value_csr_d"] = torch.from_numpy(g["value_csr"]).cuda()
    g["value_csc_d"] = torch.from_numpy(g["value_csc"]).cuda()
    g["n_v"] = n_v
    g["edge_index"] = ei
    return g


def test_spmm_function_forward_backward(pkg, graph):
    from gespmm_amd import SPMMFunction

    g = graph
    # the op's "rowptr/colind" are the CSC of adj => forward multiplies by adj^T
    At = torch.from_numpy(g["dense"].T)
    for weighted in (False, True):
        x = torch.randn(g["n_v"], 24, dtype=torch.float64)
        xd = x.float().cuda().requires_grad_(True)
        args = [g["rowptr_d"], g["colind_d"], g["colptr_d"], g["rowind_d"], xd]
        if weighted:
            args += [g["value_csr_d"], g["value_csc_d"]]
        y = SPMMFunction.apply(*args)
        assert torch.allclose(y.detach().cpu().double(), At @ x.float().double(), atol=1e-4)
        w = torch.randn_like(y)
        (y * w).sum().backward()
        assert torch.allclose(xd.grad.cpu().double(), At.T @ w.cpu().double(), atol=1e-4)
    # returns exactly the reference's gradient structure: only `feat` gets a gradient
    ew = g["value_csr_d"].clone().requires_grad_(True)
    xd = torch.randn(g["n_v"], 8, device="cuda", requires_grad=True)
    y = SPMMFunction.apply(g["rowptr_d"], g["colind_d"], g["colptr_d"], g["rowind_d"], xd, ew, g["value_csc_d"])
    y.sum().backward()
    assert ew.grad is None and xd.grad is not None


def test_spmm_function_error_behaviour(pkg, graph):
    from gespmm_amd import SPMMFunction

    g = graph
    xd = torch.randn(g["n_v"], 8, device="cuda", requires_grad=True)
    y = SPMMFunction.apply(g["rowptr_d"], g["colind_d"], g["colptr_d"], g["rowind_d"], xd, g["value_csr_d"])
    with pytest.raises(RuntimeError, match="edge values in both"):  # op.py:22-27
        y.sum().backward()


def test_edge_weight_gradient_extension(pkg, graph):
    """need_edge_grad=True: d/dw[e] = <grad_out[row(e)], feat[col(e)]> via SDDMM."""
    from gespmm_amd import SPMMFunction

    g = graph
    n = g["n_v"]
    w = torch.rand(g["colind_d"].numel(), device="cuda").requires_grad_(True)
    # CSC-ordered copy of the same weights for the backward SpMM
    from gespmm_amd import spmm

    colptr = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    rowind = torch.empty_like(g["colind_d"])
    w_csc = spmm.csr2csc(g["rowptr_d"], g["colind_d"], colptr, rowind, w.detach())
    x = torch.randn(n, 12, device="cuda", requires_grad=True)
    y = SPMMFunction.apply(g["rowptr_d"], g["colind_d"], colptr, rowind, x, w, w_csc, True)
    gout = torch.randn_like(y)
    (y * gout).sum().backward()
    rows = torch.repeat_interleave(torch.arange(n, device="cuda"),
                                   (g["rowptr_d"][1:] - g["rowptr_d"][:-1]).long())
    ref = (gout[rows] * x.detach()[g["colind_d"].long()]).sum(1)
    assert torch.allclose(w.grad, ref, atol=1e-4)
    dense = torch.zeros(n, n, device="cuda", dtype=torch.float64)
    dense[rows, g["colind_d"].long()] = w.detach().double()
    assert torch.allclose(x.grad.double(), dense.T @ gout.double(), atol=1e-4)


@pytest.mark.parametrize("tune", (False, True))
@pytest.mark.parametrize("weighted", (False, True))
def test_gcnconv_matches_dense_restatement(pkg, graph, weighted, tune):
    from gespmm_amd import GCNConv

    g = graph
    torch.manual_seed(0)
    conv = GCNConv(40, 16, cached=True, normalize=True, tune_plans=tune).cuda()  # (tune_plans: kernel choice by measurement, same results)
    with torch.no_grad():
        conv.bias.uniform_(-0.1, 0.1)
    x = torch.randn(g["n_v"], 40, device="cuda", requires_grad=True)
    args = (x, g["rowptr_d"], g["colind_d"], g["colptr_d"], g["rowind_d"])
    if weighted:
        args += (g["value_csr_d"], g["value_csc_d"])
    out = conv(*args)
    A = torch.from_numpy(g["dense"].T).cuda()  # forward operand (see above)
    in_deg = torch.from_numpy(np.diff(g["rowptr"]).astype(np.float64)).cuda()
    out_deg = torch.from_numpy(np.diff(g["colptr"]).astype(np.float64)).cuda()
    W, b = conv.weight.detach().double(), conv.bias.detach().double()
    ref = (in_deg ** -0.5).unsqueeze(1) * (A @ ((out_deg ** -0.5).unsqueeze(1) * (x.detach().double() @ W))) + b
    assert torch.allclose(out.double(), ref, atol=2e-4)
    out.pow(2).sum().backward()
    xr = x.detach().double().requires_grad_(True)
    Wr = W.clone().requires_grad_(True)
    ((in_deg ** -0.5).unsqueeze(1) * (A @ ((out_deg ** -0.5).unsqueeze(1) * (xr @ Wr))) + b).pow(2).sum().backward()
    assert torch.allclose(x.grad.double(), xr.grad, atol=2e-3)
    assert torch.allclose(conv.weight.grad.double(), Wr.grad, atol=2e-3)
    assert conv.cached_result is not None
    assert repr(conv) == "GCNConv(40, 16)"


def test_two_layer_gcn_trains(pkg, graph):
    """The caller of gcn_custom.py:63-143 in miniature: loss goes down."""
    import torch.nn.functional as F

    from gespmm_amd import GCNConv

    g = graph
    torch.manual_seed(1)
    x = torch.rand(g["n_v"], 50, device="cuda")
    x = x / x.sum(1, keepdim=True)
    ylab = torch.randint(0, 3, (g["n_v"],), device="cuda")
    c1, c2 = GCNConv(50, 32, cached=True).cuda(), GCNConv(32, 3, cached=True).cuda()
    opt = torch.optim.Adam([dict(params=c1.parameters(), weight_decay=5e-4), dict(params=c2.parameters())], lr=0.01)
    a = (g["rowptr_d"], g["colind_d"], g["colptr_d"], g["rowind_d"], g["value_csr_d"], g["value_csc_d"])
    losses = []
    for _ in range(30):
        opt.zero_grad()
        h = F.dropout(F.relu(c1(x, *a)), training=True)
        loss = F.nll_loss(F.log_softmax(c2(h, *a), dim=1), ylab)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0]


def test_spmm_plan_reuses_split_points(pkg, oracle):
    """SpmmPlan: the cache-blocked path computes its split points on the first call and reuses them;
    results equal the plan-less call bit for bit; a plan refuses another matrix; GCNConv(cached=True)
    keeps plans and trains to the same numbers as without them."""
    from gespmm_amd import GCNConv, _lib, spmm

    rng = np.random.RandomState(9)
    M = 60000
    deg = rng.randint(110, 160, size=M)  # >= 20 entries of a row per 6 MB slab: the cache-blocked path
    rowptr = np.zeros(M + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(deg)
    colind = rng.randint(0, M, size=int(rowptr[-1])).astype(np.int32)
    rp, ci = torch.from_numpy(rowptr).cuda(), torch.from_numpy(colind).cuda()
    val = torch.rand(ci.numel(), device="cuda") - 0.5
    N = 128
    assert _lib.lib.gespmm_csr_spmm_workspace_bytes(M, M, N, ci.numel(), -1, None) > 0  # dense: cache-blocked path
    plan = spmm.SpmmPlan(rp, ci, M, N)
    for it in range(3):
        B = torch.rand(M, N, device="cuda") - 0.5
        ref = spmm.csr_spmm(rp, ci, val, B)
        got = spmm.csr_spmm(rp, ci, val, B, plan=plan)
        assert torch.equal(got, ref), it
        got_u = spmm.csr_spmm_no_edge_value(rp, ci, B, plan=plan)
        assert torch.equal(got_u, spmm.csr_spmm_no_edge_value(rp, ci, B))
    with pytest.raises(ValueError):
        spmm.csr_spmm(rp, ci.clone(), val, B, plan=plan)
    # another width through the same plan is legal (scratch then comes from the library's pool)
    B64 = B[:, :64].contiguous()
    assert torch.equal(spmm.csr_spmm(rp, ci, val, B64, plan=plan), spmm.csr_spmm(rp, ci, val, B64))
    assert "order=storage" in plan.describe() and "slab-blocked" in plan.describe()
    # an in-place edit of the pattern is noticed (tensor version counters), not silently served from stale split points
    ci[0] = (ci[0] + 1) % M
    with pytest.raises(ValueError):
        spmm.csr_spmm(rp, ci, val, B, plan=plan)
