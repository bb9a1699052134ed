"""Section timings of the full-scale RMAT parity test: python scripts/time_rmat_test.py <scale>"""
import sys, time, torch, pytest
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle')
import gespmm_amd
from test_gpu_baseline_configs import _int_dense, _sub_csr, _exact_rows
def tick(msg, t=[time.time()]):
    torch.cuda.synchronize(); now=time.time()
    if now-t[0] > 0.5: print("%-62s %.2f s" % (msg, now-t[0]), flush=True)
    t[0]=now
def run(pkg, scale):
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
    tick('need = (2 * (1 << scale) * N * 4) + 16 * (1 << scale) * 12 +')
    torch.cuda.empty_cache()
    tick('torch.cuda.empty_cache()')
    free, _total = torch.cuda.mem_get_info()
    tick('free, _total = torch.cuda.mem_get_info()')
    if free < need:
        pytest.skip("scale %d needs %.0f GiB, device has %.0f GiB free" % (scale, need / 2**30, free / 2**30))
    g = graphs.rmat_shard(scale, 16, 0, 1, seed=42, device="cuda")
    tick('g = graphs.rmat_shard(scale, 16, 0, 1, seed=42, device="cuda')
    rp, ci, M, K = g["rowptr"], g["colind"], g["M"], g["K"]
    tick('rp, ci, M, K = g["rowptr"], g["colind"], g["M"], g["K"]')
    nnz = int(ci.numel())
    tick('nnz = int(ci.numel())')
    assert nnz == 16 << scale and M == K == 1 << scale
    deg = (rp[1:] - rp[:-1])
    tick('deg = (rp[1:] - rp[:-1])')
    hubs = torch.topk(deg, 8).indices.tolist()
    tick('hubs = torch.topk(deg, 8).indices.tolist()')
    assert int(deg.max()) > 2048 * 16, "the long-row pass must be in play"
    blocks = [(0, 2048), (M // 3, M // 3 + 2048), (M - 2048, M)] + [(h, h + 1) for h in hubs]
    tick('blocks = [(0, 2048), (M // 3, M // 3 + 2048), (M - 2048, M)]')

    B = _int_dense(K, N, 2654435761, "cuda")
    tick('B = _int_dense(K, N, 2654435761, "cuda")')
    gen = torch.Generator(device="cuda")
    tick('gen = torch.Generator(device="cuda")')
    gen.manual_seed(5)
    tick('gen.manual_seed(5)')
    vi = torch.randint(-2, 3, (nnz,), generator=gen, device="cuda", dtype=torch.int32).float()
    tick('vi = torch.randint(-2, 3, (nnz,), generator=gen, device="cud')
    C = torch.empty((M, N), dtype=torch.float32, device="cuda")
    tick('C = torch.empty((M, N), dtype=torch.float32, device="cuda")')
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
    tick('vf = (torch.rand(nnz, generator=gen, device="cuda") - 0.5)')
    spmm.csr_spmm(rp, ci, vf, B, out=C)
    tick('spmm.csr_spmm(rp, ci, vf, B, out=C)')
    for r0, r1 in blocks:
        rp_s, ci_s = _sub_csr(rp, ci, r0, r1)
        v_s = vf[int(rp[r0]):int(rp[r1])].double()
        rows = torch.repeat_interleave(torch.arange(r1 - r0, device="cuda"), (rp_s[1:] - rp_s[:-1]).long())
        for c0 in range(0, N, 32):
            contrib = B[ci_s.long(), c0:c0 + 32].double() * v_s.unsqueeze(1)
            if r1 - r0 == 1:
                ref, scale_abs = contrib.sum(0, keepdim=True), contrib.abs().sum(0, keepdim=True)
            else:
                ref = torch.zeros((r1 - r0, 32), dtype=torch.float64, device="cuda").index_add_(0, rows, contrib)
                scale_abs = torch.zeros_like(ref).index_add_(0, rows, contrib.abs())
            err = (C[r0:r1, c0:c0 + 32].double() - ref).abs()
            assert torch.all(err <= 1e-4 * torch.maximum(ref.abs(), scale_abs) + 1e-12), (scale, r0, c0)
            del contrib, ref, scale_abs, err
    # A . 1 = degree, every row, every column
    B.fill_(1.0)
    tick('B.fill_(1.0)')
    spmm.csr_spmm_no_edge_value(rp, ci, B, out=C)
    tick('spmm.csr_spmm_no_edge_value(rp, ci, B, out=C)')
    want = deg.float().unsqueeze(1)
    tick('want = deg.float().unsqueeze(1)')
    step = 1 << 22
    for r0 in range(0, M, step):
        assert torch.equal(C[r0:r0 + step], want[r0:r0 + step].expand(-1, N)), "A.1 != degree in rows %d.." % r0
        tick('assert torch.equal(C[r0:r0 + step], want[r0:r0 + step].expan')
    del B, C, vi, vf
    torch.cuda.empty_cache()
    tick('torch.cuda.empty_cache()')

run(gespmm_amd, int(sys.argv[1]))
