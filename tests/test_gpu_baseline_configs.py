"""One test per BASELINE.json config, at the config's own size and width, through the boundary it names.

  C1  cit-HepTh-shaped x N=32 via the spmm_test CPU-verify path   (spmm_test.cu:595-605, 671-698)
  C2b reddit-shaped x N=128                                        (the second graph of configs[1])
  C3  ogbn-products-shaped, N in {16..512}, auto-select            (configs[2])
  C4  SpMM fwd + SDDMM bwd at hidden=128 on pubmed and on the reddit-shaped graph (configs[3])
  C5  row-partitioned RMAT: shards through the HIP path on ONE device == the unsharded bits (configs[4])

Full-size graphs are checked with exact integer arithmetic (every fp32 sum is exact, so the result must EQUAL
an independent int64 computation with torch index ops) and with sampled rows against the oracle."""
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from helpers import GOLDEN, ROOT, bits

pytestmark = pytest.mark.gpu
DRIVER = os.path.join(ROOT, "gespmm_amd", "lib", "spmm_test")


def _exact_reference(rp, ci, vi, Bi, chunk=8):
    M = rp.numel() - 1
    N = Bi.shape[1]
    rows = torch.repeat_interleave(torch.arange(M, device=rp.device), (rp[1:] - rp[:-1]).long())
    out = torch.empty((M, N), dtype=torch.float32, device=rp.device)
    cil = ci.long()
    for c0 in range(0, N, chunk):
        c1 = min(c0 + chunk, N)
        contrib = Bi[cil, c0:c1].long()
        if vi is not None:
            contrib = contrib * vi.long().unsqueeze(1)
        ref = torch.zeros((M, c1 - c0), dtype=torch.int64, device=rp.device)
        ref.index_add_(0, rows, contrib)
        out[:, c0:c1] = ref.float()
        del contrib, ref
    return out


def _sampled_rows_vs_oracle(oracle, rp, ci, val, B, C, nrows=256, seed=0, mode="fma"):
    M = rp.numel() - 1
    rng = np.random.RandomState(seed)
    rows = np.sort(rng.choice(M, min(nrows, M), replace=False))
    rph, cih = rp.cpu().numpy(), ci.cpu().numpy()
    sub_ptr = np.zeros(len(rows) + 1, dtype=np.int32)
    sub_ptr[1:] = np.cumsum(rph[rows + 1] - rph[rows])
    sel = np.concatenate([np.arange(rph[r], rph[r + 1]) for r in rows]).astype(np.int64)
    vh = val.cpu().numpy()[sel] if val is not None else None
    cols_u, inv = np.unique(cih[sel], return_inverse=True)
    Bsub = B[torch.from_numpy(cols_u.astype(np.int64)).to(B.device)].cpu().numpy()
    ref = oracle.spmm(sub_ptr, inv.astype(np.int32), vh, Bsub, mode)
    got = C[torch.from_numpy(rows).to(C.device)].cpu().numpy()
    return bool(np.array_equal(bits(got), bits(ref)))


# ----------------------------------------------------------------------------- C1

def test_c1_cit_hepth_n32_driver_cpu_verify(pkg, tmp_path):
    """BASELINE configs[0]: the driver's own CPU-verify path at N=32 on the cit-HepTh-shaped stand-in written
    as a .mtx file, every variant + the library's pick, CPU loop timed beside it, atomic baseline column."""
    from gespmm_amd import graphs

    g = graphs.synthetic_graph("cit-hepth-like", seed=42, device="cpu")
    assert g["M"] == 27770 and g["nnz"] == 352807
    mtx = tmp_path / "cit-hepth-like.mtx"
    graphs.write_mtx(str(mtx), g["rowptr"], g["colind"])
    r = subprocess.run([DRIVER, str(mtx), "0", "--ncols", "32", "--method", "-1", "--validate", "--cpu-baseline",
                        "--atomic-baseline", "--iters", "50", "--seed", "1", "--describe", "--out",
                        str(tmp_path / "o.csv")], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "read file ok. N=27770 nnz=352807" in r.stdout
    assert "WA" not in r.stdout, r.stdout
    assert "validate done (7 variants, N=32)" in r.stdout  # methods 0..5 and the AUTO pick
    assert re.search(r"cpu golden loop: [0-9.]+ GFLOP/s \(1 thread, N=32\)", r.stdout)
    assert re.search(r"N=32 launches: variant=1 kernel=batch-stream", r.stdout), r.stdout
    m = re.search(r"N=32 method=-1: [0-9.]+ ms/iter, ([0-9.]+) GFLOP/s", r.stdout)
    assert m and float(m.group(1)) > 100.0
    assert re.search(r"N=32 atomic-baseline: [0-9.]+ ms/iter, [0-9.]+ GFLOP/s", r.stdout)


def test_c1_cit_hepth_n32_bits(pkg, oracle):
    """The same config through the C ABI: all bit-exact variants equal the oracle's golden loop on the whole matrix."""
    from gespmm_amd import graphs, spmm

    g = graphs.synthetic_graph("cit-hepth-like", seed=42, device="cpu")
    rp, ci = g["rowptr"].numpy(), g["colind"].numpy()
    B = oracle.hash_B(g["K"], 32, seed=1)
    ref = oracle.spmm(rp, ci, None, B, mode="golden")
    rpd, cid, Bd = g["rowptr"].cuda(), g["colind"].cuda(), torch.from_numpy(B).cuda()
    for variant in (-1, 0, 1, 2, 3, 4):
        C = spmm.csr_spmm_no_edge_value(rpd, cid, Bd, variant=variant).cpu().numpy()
        assert np.array_equal(bits(C), bits(ref)), variant


# ----------------------------------------------------------------------------- C2b / C4-reddit

@pytest.fixture(scope="module")
def reddit(pkg):
    from gespmm_amd import graphs

    g = graphs.synthetic_graph("reddit-like", seed=42, device="cuda")
    assert g["M"] == 232965 and g["nnz"] == 114615892
    yield g
    torch.cuda.empty_cache()


def test_c2b_reddit_n128(pkg, oracle, reddit):
    from gespmm_amd import _lib, spmm

    rp, ci, M = reddit["rowptr"], reddit["colind"], reddit["M"]
    what = ctypes_describe(M, M, 128, reddit["nnz"])
    assert "kernel=slab-blocked" in what, what
    gen = torch.Generator(device="cuda")
    gen.manual_seed(128)
    vi = torch.randint(-2, 3, (reddit["nnz"],), generator=gen, device="cuda", dtype=torch.int32)
    Bi = torch.randint(-4, 5, (M, 128), generator=gen, device="cuda", dtype=torch.int32)
    ref = _exact_reference(rp, ci, vi, Bi)
    C = spmm.csr_spmm(rp, ci, vi.float(), Bi.float())
    assert torch.equal(C, ref), "AUTO (cache-blocked) N=128"
    plan = spmm.SpmmPlan(rp, ci, M, 128)
    for _ in range(2):  # second call reuses the split points
        assert torch.equal(spmm.csr_spmm(rp, ci, vi.float(), Bi.float(), plan=plan), ref)
    del ref, C
    # real-valued operands: sampled rows against the oracle's device-arithmetic chain, bit for bit
    val = torch.rand(reddit["nnz"], generator=gen, device="cuda") - 0.5
    B = (torch.randint(0, 100, (M, 128), generator=gen, device="cuda", dtype=torch.int32) - 50).float() / 100
    C = spmm.csr_spmm(rp, ci, val, B)
    assert _sampled_rows_vs_oracle(oracle, rp, ci, val, B, C, nrows=128)


def ctypes_describe(M, K, N, nnz, variant=-1):
    import ctypes

    from gespmm_amd import _lib

    buf = ctypes.create_string_buffer(256)
    n = _lib.lib.gespmm_describe_launch(M, K, N, nnz, variant, None, buf, 256)
    assert n > 0
    return buf.value.decode()


def test_c4_reddit_spmm_fwd_sddmm_bwd(pkg, oracle, reddit):
    """One aggregation step of config 4 on the reddit-shaped graph at hidden=128: SPMMFunction forward,
    backward to the features (SpMM on the CSC arrays) and to the edge weights (SDDMM)."""
    import gespmm_amd
    from gespmm_amd import graphs

    rp, ci, M, nnz = reddit["rowptr"], reddit["colind"], reddit["M"], reddit["nnz"]
    gen = torch.Generator(device="cuda")
    gen.manual_seed(4)
    w = (torch.rand(nnz, generator=gen, device="cuda") - 0.5).requires_grad_(True)
    colptr, rowind, w_csc = graphs.transpose_csr(rp, ci, val=w.detach())
    x = ((torch.randint(0, 100, (M, 128), generator=gen, device="cuda", dtype=torch.int32) - 50).float() / 100)
    x.requires_grad_(True)
    y = gespmm_amd.SPMMFunction.apply(rp, ci, colptr, rowind, x, w, w_csc, True)
    go = ((torch.randint(0, 100, (M, 128), generator=gen, device="cuda", dtype=torch.int32) - 50).float() / 100)
    y.backward(go)
    assert _sampled_rows_vs_oracle(oracle, rp, ci, w.detach(), x.detach(), y.detach(), nrows=64)
    assert _sampled_rows_vs_oracle(oracle, colptr, rowind, w_csc, go, x.grad, nrows=64)
    # edge-weight gradient: <go[r_e], x[c_e]>, sampled edges in float64
    e = torch.randint(0, nnz, (4096,), generator=gen, device="cuda")
    rows = torch.searchsorted(rp.long(), e, right=True) - 1
    ref = (go[rows].double() * x.detach()[ci[e].long()].double()).sum(1)
    scale = (go[rows].double().abs() * x.detach()[ci[e].long()].double().abs()).sum(1)
    assert torch.all((w.grad[e].double() - ref).abs() <= 1e-4 * torch.maximum(ref.abs(), scale) + 1e-12)


# ----------------------------------------------------------------------------- C3

def test_c3_products_width_sweep_auto_select(pkg, oracle):
    """configs[2]: N in {16,32,64,128,256,512} on the ogbn-products-shaped graph with the library's own choice;
    the choice itself is pinned (CRC up to 64 columns, CRC+CWM4 beyond, streaming kernel — never cache-blocked)."""
    from gespmm_amd import graphs, spmm

    g = graphs.synthetic_graph("products-like", seed=42, device="cuda")
    assert g["M"] == 2449029 and g["nnz"] == 123718280
    rp, ci, M, nnz = g["rowptr"], g["colind"], g["M"], g["nnz"]
    gen = torch.Generator(device="cuda")
    gen.manual_seed(3)
    vi = torch.randint(-2, 3, (nnz,), generator=gen, device="cuda", dtype=torch.int32)
    expect = {16: "variant=1 kernel=batch-stream V=1 S=1 W=16", 32: "variant=1 kernel=batch-stream V=1 S=1 W=32",
              64: "variant=1 kernel=batch-stream V=1 S=1 W=64", 128: "variant=3 kernel=batch-stream V=4 S=1 W=32",
              256: "variant=3 kernel=batch-stream V=4 S=1 W=64", 512: "variant=3 kernel=batch-stream V=4 S=1 W=64"}
    for N in (16, 32, 64, 128, 256, 512):
        what = ctypes_describe(M, M, N, nnz)
        assert what.startswith(expect[N]), (N, what)
        Bi = torch.randint(-4, 5, (M, N), generator=gen, device="cuda", dtype=torch.int32)
        C = spmm.csr_spmm(rp, ci, vi.float(), Bi.float())
        # exact check in column chunks without holding a second M x N matrix
        rows = torch.repeat_interleave(torch.arange(M, device="cuda"), (rp[1:] - rp[:-1]).long())
        cil = ci.long()
        step = 8
        for c0 in range(0, N, step):
            c1 = min(c0 + step, N)
            ref = torch.zeros((M, c1 - c0), dtype=torch.int64, device="cuda")
            ref.index_add_(0, rows, Bi[cil, c0:c1].long() * vi.long().unsqueeze(1))
            assert torch.equal(C[:, c0:c1].long(), ref), (N, c0)
            del ref
        del Bi, C, rows, cil
        torch.cuda.empty_cache()


# ----------------------------------------------------------------------------- C4 pubmed

def test_c4_pubmed_hidden128_fwd_bwd_vs_oracle(pkg, oracle, bundled):
    """pubmed (+ self loops) at hidden=128: forward bits == oracle chain, feature gradient bits == oracle chain on
    the transposed pattern, edge-weight gradient (SDDMM) within 1e-4 of the float64 oracle."""
    import gespmm_amd
    from gespmm_amd import graphs

    g = bundled["pubmed"]
    rp0, ci0 = torch.from_numpy(g["rowptr"]).cuda(), torch.from_numpy(g["colind"]).cuda()
    rp, ci = graphs.add_self_loops(rp0, ci0)
    M, nnz = g["M"], int(ci.numel())
    assert nnz == 88648 + 19717
    w_h = oracle.hash_val(nnz, seed=5)
    w = torch.from_numpy(w_h).cuda().requires_grad_(True)
    colptr, rowind, w_csc = graphs.transpose_csr(rp, ci, val=w.detach())
    x_h = oracle.hash_B(M, 128, seed=2)
    go_h = oracle.hash_B(M, 128, seed=3)
    x = torch.from_numpy(x_h).cuda().requires_grad_(True)
    y = gespmm_amd.SPMMFunction.apply(rp, ci, colptr, rowind, x, w, w_csc, True)
    y.backward(torch.from_numpy(go_h).cuda())
    rph, cih = rp.cpu().numpy(), ci.cpu().numpy()
    ref = oracle.spmm(rph, cih, w_h, x_h, "fma")
    assert np.array_equal(bits(y.detach().cpu().numpy()), bits(ref))
    refg = oracle.spmm(colptr.cpu().numpy(), rowind.cpu().numpy(), w_csc.cpu().numpy(), go_h, "fma")
    assert np.array_equal(bits(x.grad.cpu().numpy()), bits(refg))
    ref_e, scale_e = oracle.sddmm(rph, cih, go_h, x_h, csr=True)
    got = w.grad.cpu().numpy().astype(np.float64)
    assert np.all(np.abs(got - ref_e.astype(np.float64)) <= 1e-4 * np.maximum(np.abs(ref_e), scale_e) + 1e-12)


# ----------------------------------------------------------------------------- C5 (single-GPU shard emulation)

@pytest.mark.parametrize("graph", ("pubmed", "rmat20"))
def test_c5_row_shards_through_hip_equal_unsharded_bits(pkg, bundled, graph):
    """Partition with gespmm_row_partition into 2/4/8 shards, run EVERY shard through the HIP path on this
    device with the full B, concatenate: must equal the unsharded launch bit for bit (strict order: the
    long-row pass re-associates per launch, so it is pinned off on both sides for the RMAT graph)."""
    from gespmm_amd import _lib, graphs, spmm

    if graph == "pubmed":
        g = bundled["pubmed"]
        rp, ci = torch.from_numpy(g["rowptr"]).cuda(), torch.from_numpy(g["colind"]).cuda()
        K, N = g["K"], 128
    else:
        g = graphs.rmat_shard(20, 16, 0, 1, seed=42, device="cuda")
        rp, ci, K, N = g["rowptr"], g["colind"], g["K"], 256
    M, nnz = rp.numel() - 1, int(ci.numel())
    gen = torch.Generator(device="cuda")
    gen.manual_seed(11)
    val = torch.rand(nnz, generator=gen, device="cuda") - 0.5
    B = (torch.randint(0, 100, (K, N), generator=gen, device="cuda", dtype=torch.int32) - 50).float() / 100
    cfg = {"flags": _lib.FLAG_STRICT_ORDER}
    whole = spmm.csr_spmm(rp, ci, val, B, cfg=cfg)
    rph = rp.cpu().numpy()
    for parts in (2, 4, 8):
        cut = graphs.row_partition(rph, parts)
        assert cut[0] == 0 and cut[-1] == M and np.all(np.diff(cut) >= 0)
        pieces = []
        for p in range(parts):
            r0, r1 = int(cut[p]), int(cut[p + 1])
            e0, e1 = int(rph[r0]), int(rph[r1])
            rp_loc = (rp[r0:r1 + 1] - e0).contiguous()
            ci_loc = ci[e0:e1].clone()      # fresh allocations: a shard is its own matrix on its own rank
            val_loc = val[e0:e1].clone()
            pieces.append(spmm.csr_spmm(rp_loc, ci_loc, val_loc, B, cfg=cfg))
            # shard balance: no shard holds more than its share plus one row's worth
            assert e1 - e0 <= nnz // parts + int((rp[1:] - rp[:-1]).max()) + 1
        got = torch.cat(pieces, 0)
        assert torch.equal(got.view(torch.int32), whole.view(torch.int32)), (graph, parts)
    # default flags (long-row pass on for the RMAT graph): shards agree with the whole within the 1e-4 bar
    whole_d = spmm.csr_spmm(rp, ci, val, B)
    scale = spmm.csr_spmm(rp, ci, val.abs(), B.abs(), cfg=cfg)
    assert torch.all((whole_d - whole).abs() <= 1e-4 * torch.maximum(whole.abs(), scale) + 1e-12)


# ----------------------------------------------------------------------------- C5 at its own size (one GPU)

def _int_dense(K, N, mult, device):
    """Deterministic integer-valued B in [-4, 4], filled in row chunks (no K x N temporaries)."""
    B = torch.empty((K, N), dtype=torch.float32, device=device)
    j = torch.arange(N, device=device, dtype=torch.int64).unsqueeze(0)
    step = 1 << 20
    for k0 in range(0, K, step):
        k = torch.arange(k0, min(k0 + step, K), device=device, dtype=torch.int64).unsqueeze(1)
        B[k0:k0 + step] = ((((k * mult + j * 40503 + 12345) >> 7) % 9) - 4).float()
    return B


def _sub_csr(rp, ci, r0, r1):
    e0, e1 = int(rp[r0]), int(rp[r1])
    return (rp[r0:r1 + 1] - e0).contiguous(), ci[e0:e1]


def _exact_rows(rp_s, ci_s, v_s, B, chunk=8):
    """int64 index_add over a row block of A against the full B, column chunks (integer inputs)."""
    m = rp_s.numel() - 1
    N = B.shape[1]
    rows = torch.repeat_interleave(torch.arange(m, device=B.device), (rp_s[1:] - rp_s[:-1]).long())
    cil = ci_s.long()
    out = torch.empty((m, N), dtype=torch.float32, device=B.device)
    for c0 in range(0, N, chunk):
        contrib = B[cil, c0:c0 + chunk].long()
        if v_s is not None:
            contrib = contrib * v_s.long().unsqueeze(1)
        if m == 1:  # one (hub) row: a plain sum — a million atomics on one address crawl
            acc = contrib.sum(0, keepdim=True)
        else:
            acc = torch.zeros((m, contrib.shape[1]), dtype=torch.int64, device=B.device)
            acc.index_add_(0, rows, contrib)
        out[:, c0:c0 + chunk] = acc.float()
        del contrib, acc
    return out


@pytest.mark.parametrize("scale", (24, 26))
def test_c5_rmat_at_full_scale_n256(pkg, scale):
    """BASELINE configs[4] at its own size on ONE device: RMAT scale 26 (2^30 entries, B and C 64 GiB each)
    x N = 256 when the device has the memory (MI355X: 288 GB), scale 24 always. 64-bit offsets into B
    (K*N = 2^34), the long-row pass on (hub rows of 10^5..10^6 entries). Size-independent properties:
      * integer-valued A and B make every association exact -> sampled row blocks AND the longest rows
        must equal an independent int64 computation bit for bit (valued and unweighted);
      * A . 1 = row degree for every row (unweighted, all columns);
      * float values: the longest rows within 1e-4 * sum|a.b| of a float64 reference (north_star's bar,
        scaled as SURVEY.md section 8 c4 prescribes), short rows bit-exact against the fma chain in CSR order."""
    from gespmm_amd import graphs, spmm

    N = 256
    need = (2 * (1 << scale) * N * 4) + 16 * (1 << scale) * 12 + (8 << 30)
    torch.cuda.empty_cache()
    free, _total = torch.cuda.mem_get_info()
    if free < need:
        pytest.skip("scale %d needs %.0f GiB, device has %.0f GiB free" % (scale, need / 2**30, free / 2**30))
    g = graphs.rmat_shard(scale, 16, 0, 1, seed=42, device="cuda")
    rp, ci, M, K = g["rowptr"], g["colind"], g["M"], g["K"]
    nnz = int(ci.numel())
    assert nnz == 16 << scale and M == K == 1 << scale
    deg = (rp[1:] - rp[:-1])
    hubs = torch.topk(deg, 8).indices.tolist()
    assert int(deg.max()) > 2048 * 16, "the long-row pass must be in play"
    blocks = [(0, 2048), (M // 3, M // 3 + 2048), (M - 2048, M)] + [(h, h + 1) for h in hubs]

    B = _int_dense(K, N, 2654435761, "cuda")
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    vi = torch.randint(-2, 3, (nnz,), generator=gen, device="cuda", dtype=torch.int32).float()
    C = torch.empty((M, N), dtype=torch.float32, device="cuda")
    # unweighted, then integer-valued
    for v in (None, vi):
        if v is None:
            spmm.csr_spmm_no_edge_value(rp, ci, B, out=C)
        else:
            spmm.csr_spmm(rp, ci, v, B, out=C)
        for r0, r1 in blocks:
            rp_s, ci_s = _sub_csr(rp, ci, r0, r1)
            v_s = None if v is None else v[int(rp[r0]):int(rp[r1])]
            ref = _exact_rows(rp_s, ci_s, v_s, B)
            assert torch.equal(C[r0:r1], ref), ("scale %d rows %d..%d valued=%s" % (scale, r0, r1, v is not None))
            del ref
    # float values: hubs by tolerance, a short-row block bit-exact vs a sequential fp32 fma chain is covered
    # at small sizes; here: |C - float64 reference| <= 1e-4 * sum|a.b| on the sampled rows
    vf = (torch.rand(nnz, generator=gen, device="cuda") - 0.5)
    spmm.csr_spmm(rp, ci, vf, B, out=C)
    for r0, r1 in blocks:
        rp_s, ci_s = _sub_csr(rp, ci, r0, r1)
        v_s = vf[int(rp[r0]):int(rp[r1])].double()
        for c0 in range(0, N, 32):
            contrib = B[ci_s.long(), c0:c0 + 32].double() * v_s.unsqueeze(1)
            # segment sums by prefix-sum differences (float64 index_add_ crawls on this stack)
            ends = rp_s[1:].long()
            starts = rp_s[:-1].long()
            zero = torch.zeros((1, contrib.shape[1]), dtype=torch.float64, device="cuda")
            cs = torch.cat([zero, contrib.cumsum(0)])
            ca = torch.cat([zero, contrib.abs().cumsum(0)])
            ref, scale_abs = cs[ends] - cs[starts], ca[ends] - ca[starts]
            del cs, ca
            err = (C[r0:r1, c0:c0 + 32].double() - ref).abs()
            assert torch.all(err <= 1e-4 * torch.maximum(ref.abs(), scale_abs) + 1e-9), (scale, r0, c0)
            del contrib, ref, scale_abs, err
    # A . 1 = degree, every row, every column
    B.fill_(1.0)
    spmm.csr_spmm_no_edge_value(rp, ci, B, out=C)
    want = deg.float().unsqueeze(1)
    step = 1 << 22
    for r0 in range(0, M, step):
        assert torch.equal(C[r0:r0 + step], want[r0:r0 + step].expand(-1, N)), "A.1 != degree in rows %d.." % r0
    del B, C, vi, vf
    torch.cuda.empty_cache()
