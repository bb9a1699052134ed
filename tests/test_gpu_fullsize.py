"""BASELINE.json's full-size workload (com-Amazon-like, N=128), checked through
size-independent properties instead of the (slow) oracle:
  * integer-valued inputs make every fp32 sum exact, so the result must EQUAL an
    independent int64 computation (torch index_select / index_add on the GPU);
  * B == 1 gives the row degrees;
  * linearity in B within fp32 rounding;
  * a sampled set of rows against the oracle, bit for bit."""
import numpy as np
import pytest
import torch

from helpers import bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amazon(pkg):
    from gespmm_amd import graphs

    g = graphs.synthetic_graph("com-amazon-like", seed=42, device="cuda")
    assert g["M"] == 334863 and g["nnz"] == 1851744
    return g


def test_generator_contract(amazon):
    rp, ci = amazon["rowptr"].long(), amazon["colind"].long()
    M = amazon["M"]
    rows = torch.repeat_interleave(torch.arange(M, device="cuda"), rp[1:] - rp[:-1])
    key = rows * M + ci
    assert torch.all(key[1:] > key[:-1]), "sorted, no duplicates"
    assert not torch.any(rows == ci), "no self loops"
    tkey, _ = torch.sort(ci * M + rows)
    assert torch.equal(tkey, key), "symmetric"
    deg = (rp[1:] - rp[:-1])
    assert 100 < int(deg.max()) < 2000


@pytest.mark.parametrize("N", (32, 128, 512))
def test_exact_integer_arithmetic_full_size(pkg, amazon, N):
    from gespmm_amd import spmm

    M = amazon["M"]
    gen = torch.Generator(device="cuda")
    gen.manual_seed(N)
    Bi = torch.randint(-8, 9, (M, N), generator=gen, device="cuda", dtype=torch.int32)
    vi = torch.randint(-4, 5, (amazon["nnz"],), generator=gen, device="cuda", dtype=torch.int32)
    rp, ci = amazon["rowptr"], amazon["colind"]
    rows = torch.repeat_interleave(torch.arange(M, device="cuda"), (rp[1:] - rp[:-1]).long())
    # independent exact reference in int64, chunked over columns to bound memory
    for variant in (-1, 2, 4):
        C = spmm.csr_spmm(rp, ci, vi.float(), Bi.float(), variant=variant)
        for c0 in range(0, N, 64):
            c1 = min(c0 + 64, N)
            contrib = Bi[ci.long(), c0:c1].long() * vi.long().unsqueeze(1)
            ref = torch.zeros((M, c1 - c0), dtype=torch.int64, device="cuda")
            ref.index_add_(0, rows, contrib)
            assert torch.equal(C[:, c0:c1].long(), ref), (N, variant, c0)
    Cu = spmm.csr_spmm_no_edge_value(rp, ci, torch.ones(M, N, device="cuda"))
    deg = (rp[1:] - rp[:-1]).float().unsqueeze(1).expand(M, N)
    assert torch.equal(Cu, deg), "A @ 1 = row degree"


def test_linearity_and_sampled_oracle_rows(pkg, oracle, amazon):
    from gespmm_amd import spmm

    M, N = amazon["M"], 128
    rp, ci = amazon["rowptr"], amazon["colind"]
    B1 = torch.from_numpy(oracle.hash_B(M, N, seed=1)).cuda()
    B2 = torch.from_numpy(oracle.hash_B(M, N, seed=2)).cuda()
    val = torch.from_numpy(oracle.hash_val(amazon["nnz"], seed=7)).cuda()
    C1 = spmm.csr_spmm(rp, ci, val, B1)
    C2 = spmm.csr_spmm(rp, ci, val, B2)
    C12 = spmm.csr_spmm(rp, ci, val, B1 + B2)
    scale = spmm.csr_spmm(rp, ci, val.abs(), B1.abs() + B2.abs())
    assert torch.all((C12 - (C1 + C2)).abs() <= 1e-5 * scale + 1e-12)
    # 2000 sampled rows, bit-exact against the oracle on the extracted sub-matrix
    rng = np.random.RandomState(0)
    rows = np.sort(rng.choice(M, 2000, replace=False))
    rph, cih, vh = rp.cpu().numpy(), ci.cpu().numpy(), val.cpu().numpy()
    sub_ptr = np.zeros(len(rows) + 1, dtype=np.int32)
    sub_ptr[1:] = np.cumsum(rph[rows + 1] - rph[rows])
    sel = np.concatenate([np.arange(rph[r], rph[r + 1]) for r in rows])
    ref = oracle.spmm(sub_ptr, cih[sel], vh[sel], B1.cpu().numpy(), "fma")
    got = C1[torch.from_numpy(rows).cuda()].cpu().numpy()
    assert np.array_equal(bits(got), bits(ref))


def test_offsets_beyond_4GB_of_B(pkg):
    """K*N*4 >= 2^32 switches the kernels to 64-bit byte offsets (RMAT-26 x N=256 needs
    2^34 elements; reference: 32-bit `col*N`, spmm_test.cu:124, overflows). Rows that
    reach past the 4 GiB mark must gather the right B rows — checked with exact integer
    arithmetic against torch index ops."""
    from gespmm_amd import spmm

    N, K, M = 512, (1 << 21) + 37, 3000  # B = 4.3 GB
    assert K * N * 4 >= (1 << 32)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    B = torch.randint(-8, 9, (K, N), generator=gen, device="cuda", dtype=torch.int32).float()
    deg = torch.randint(0, 40, (M,), generator=gen, device="cuda")
    deg[7] = 3000  # one long row
    rowptr = torch.zeros(M + 1, dtype=torch.int64, device="cuda")
    rowptr[1:] = torch.cumsum(deg, 0)
    nnz = int(rowptr[-1])
    # half the entries point into the last 5 % of B (beyond 4 GiB), including the very last row
    hi = torch.randint(int(K * 0.95), K, (nnz,), generator=gen, device="cuda")
    lo = torch.randint(0, K, (nnz,), generator=gen, device="cuda")
    col = torch.where(torch.rand(nnz, generator=gen, device="cuda") < 0.5, hi, lo)
    col[0] = K - 1
    val = torch.randint(-3, 4, (nnz,), generator=gen, device="cuda").float()
    rp32, ci32 = rowptr.to(torch.int32), col.to(torch.int32)
    rows = torch.repeat_interleave(torch.arange(M, device="cuda"), deg)
    ref = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    ref.index_add_(0, rows, B[col] * val.unsqueeze(1))  # small integers: exact in fp32
    for variant in (-1, 0, 1, 2, 4, 5):
        C = spmm.csr_spmm(rp32, ci32, val, B, variant=variant)
        assert torch.equal(C, ref), "variant %d" % variant
    Cu = spmm.csr_spmm_no_edge_value(rp32, ci32, B)
    refu = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    refu.index_add_(0, rows, B[col])
    assert torch.equal(Cu, refu)
    del B, ref, refu
    torch.cuda.empty_cache()


def test_wide_and_odd_feature_widths_full_rows(pkg, oracle):
    """N far beyond one column tile (N = 2048, 1000, 1023): many column tiles per row."""
    from gespmm_amd import graphs, spmm

    g = graphs.synthetic_graph("cit-hepth-like", seed=3, device="cuda")
    rp, ci = g["rowptr"], g["colind"]
    M = g["M"]
    rows = torch.repeat_interleave(torch.arange(M, device="cuda"), (rp[1:] - rp[:-1]).long())
    for N in (1000, 1023, 2048):
        gen = torch.Generator(device="cuda")
        gen.manual_seed(N)
        B = torch.randint(-8, 9, (M, N), generator=gen, device="cuda", dtype=torch.int32).float()
        ref = torch.zeros((M, N), dtype=torch.float32, device="cuda")
        ref.index_add_(0, rows, B[ci.long()])
        for variant in (-1, 1, 3, 4):
            assert torch.equal(spmm.csr_spmm_no_edge_value(rp, ci, B, variant=variant), ref), (N, variant)


def _exact_reference(rp, ci, vi, Bi, chunk=8):
    """Independent exact result for integer-valued inputs: int64 index_add, column chunks."""
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


def test_dense_graph_auto_path_full_size(pkg):
    """reddit-like (115 M nnz): AUTO takes the cache-blocked (slab) path at N=64 and the
    streaming + long-row path at N=16; both must reproduce exact integer arithmetic, and
    the max reducer (long rows split across a workgroup) must equal torch's segment max."""
    from gespmm_amd import _lib, graphs, spmm

    g = graphs.synthetic_graph("reddit-like", seed=42, device="cuda")
    rp, ci, M = g["rowptr"], g["colind"], g["M"]
    gen = torch.Generator(device="cuda")
    gen.manual_seed(3)
    vi = torch.randint(-2, 3, (g["nnz"],), generator=gen, device="cuda", dtype=torch.int32)
    for N in (64, 16):
        Bi = torch.randint(-4, 5, (M, N), generator=gen, device="cuda", dtype=torch.int32)
        ref = _exact_reference(rp, ci, vi, Bi)
        C = spmm.csr_spmm(rp, ci, vi.float(), Bi.float())
        assert torch.equal(C, ref), "auto path N=%d" % N
        Cs = spmm.csr_spmm(rp, ci, vi.float(), Bi.float(),
                           cfg={"flags": _lib.FLAG_NO_SLAB_BLOCKED | _lib.FLAG_STRICT_ORDER})
        assert torch.equal(Cs, ref), "streaming strict N=%d" % N
        del ref, C, Cs
    B = torch.rand(M, 16, generator=gen, device="cuda")
    rows = torch.repeat_interleave(torch.arange(M, device="cuda"), (rp[1:] - rp[:-1]).long())
    refm = torch.full((M, 16), -10000.0, device="cuda")
    refm.scatter_reduce_(0, rows.unsqueeze(1).expand(-1, 16), B[ci.long()], reduce="amax", include_self=True)
    assert torch.equal(spmm.csr_spmm_max(rp, ci, B), refm)


def test_rmat_long_rows_full_size(pkg):
    """RMAT scale 20 (16.8 M nnz, hub rows of ~10^5 entries): the long-row pass is on
    (nnz >= 2^23); integer-valued inputs make every association exact."""
    from gespmm_amd import graphs, spmm

    g = graphs.rmat_shard(20, 16, 0, 1, seed=42, device="cuda")
    rp, ci, M = g["rowptr"], g["colind"], g["M"]
    deg = rp[1:] - rp[:-1]
    assert int(deg.max()) > 20000
    gen = torch.Generator(device="cuda")
    gen.manual_seed(9)
    for N in (128, 32):
        Bi = torch.randint(-4, 5, (g["K"], N), generator=gen, device="cuda", dtype=torch.int32)
        ref = _exact_reference(rp, ci, None, Bi, chunk=16)
        assert torch.equal(spmm.csr_spmm_no_edge_value(rp, ci, Bi.float()), ref), N
