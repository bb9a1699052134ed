"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the
same inputs. Variants 0-4 must be BIT-EXACT (integer compare of the fp32 bit patterns)
against the device-arithmetic restatement, and — unweighted — against the reference's
CPU golden loop; the parallel-reduction variant is held to north_star's 1e-4 relative
tolerance, scaled by sum|a*b| as SURVEY.md §8(c4) prescribes."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, bits, edge_case_csr

pytestmark = pytest.mark.gpu

EXACT_VARIANTS = (-1, 0, 1, 2, 3, 4)
N_SWEEP = (1, 2, 3, 4, 5, 8, 16, 31, 32, 33, 41, 64, 65, 100, 127, 128, 130, 192, 256, 258, 512)


def dev_csr(G, dev="cuda"):
    return (torch.from_numpy(np.ascontiguousarray(G["rowptr"])).to(dev),
            torch.from_numpy(np.ascontiguousarray(G["colind"])).to(dev))


def run(pkg, G, B, val=None, variant=-1, cfg=None):
    from gespmm_amd import spmm

    rp, ci = dev_csr(G)
    Bd = torch.from_numpy(B).cuda()
    if val is None:
        C = spmm.csr_spmm_no_edge_value(rp, ci, Bd, variant=variant, cfg=cfg)
    else:
        C = spmm.csr_spmm(rp, ci, torch.from_numpy(val).cuda(), Bd, variant=variant, cfg=cfg)
    torch.cuda.synchronize()
    return C.cpu().numpy()


def assert_bits_equal(a, b, what):
    if not np.array_equal(bits(a), bits(b)):
        bad = np.argwhere(bits(a) != bits(b))
        r, c = bad[0]
        raise AssertionError("%s: %d/%d elements differ, first at [%d,%d]: %r vs %r" %
                             (what, len(bad), a.size, r, c, a[r, c], b[r, c]))


@pytest.mark.parametrize("g", ("cora", "citeseer", "pubmed"))
def test_bundled_graphs_bit_exact_all_variants(pkg, oracle, bundled, g):
    G = bundled[g]
    val = oracle.hash_val(G["nnz"], seed=7)
    for N in (3, 16, 32, 41, 64, 128, 512):
        if g == "pubmed" and N == 512:
            continue
        B = oracle.hash_B(G["K"], N, seed=1)
        ref_u = oracle.spmm(G["rowptr"], G["colind"], None, B, "golden")  # reference CPU golden
        ref_v = oracle.spmm(G["rowptr"], G["colind"], val, B, "fma")      # reference device arithmetic
        for variant in EXACT_VARIANTS:
            assert_bits_equal(run(pkg, G, B, None, variant), ref_u, "%s N=%d unweighted v%d" % (g, N, variant))
            assert_bits_equal(run(pkg, G, B, val, variant), ref_v, "%s N=%d valued v%d" % (g, N, variant))


def test_committed_golden_vectors(pkg, oracle, bundled):
    """HIP output against the vectors committed under tests/golden/ (no oracle call)."""
    with open(os.path.join(GOLDEN, "spmm_checksums.json")) as f:
        sums = json.load(f)["graphs"]
    for g in ("cora", "citeseer", "pubmed"):
        G = bundled[g]
        val = oracle.hash_val(G["nnz"], seed=7)
        for N in (3, 16, 32, 41, 64, 128, 512):
            B = oracle.hash_B(G["K"], N, seed=1)
            for mode, v in (("unweighted_golden", None), ("valued_fma", val)):
                C = run(pkg, G, B, v)
                exp = sums[g][str(N)][mode]
                assert int(np.bitwise_xor.reduce(bits(C).ravel())) == exp["xor"], (g, N, mode)
                assert float(C.astype(np.float64).sum()) == exp["sum"], (g, N, mode)
                for r, c, b in exp["samples"]:
                    assert int(bits(C[r, c:c + 1])[0]) == b


@pytest.mark.parametrize("seed", (0, 1))
def test_edge_shapes_every_width(pkg, oracle, seed):
    """Empty rows, rows of 63/64/65/127/128/129/200 entries, ragged M, K != M, unsorted
    and repeated column indices — across N that are not multiples of any tile."""
    G = edge_case_csr(seed)
    val = oracle.hash_val(G["nnz"], seed=3)
    for N in N_SWEEP:
        B = oracle.hash_B(G["K"], N, seed=10 + N)
        ref_u = oracle.spmm(G["rowptr"], G["colind"], None, B, "golden")
        ref_v = oracle.spmm(G["rowptr"], G["colind"], val, B, "fma")
        for variant in EXACT_VARIANTS:
            assert_bits_equal(run(pkg, G, B, None, variant), ref_u, "edge N=%d unweighted v%d" % (N, variant))
            assert_bits_equal(run(pkg, G, B, val, variant), ref_v, "edge N=%d valued v%d" % (N, variant))
        from gespmm_amd import _lib

        for rpw in (1, 2, 3, 8, 21, 32):  # rows per lane group / per wavefront, incl. > M
            for variant in (1, 3, 4):
                for flags in (0, _lib.FLAG_BATCH_STREAM):
                    cfg = {"rows_per_wave": rpw, "flags": flags}
                    assert_bits_equal(run(pkg, G, B, val, variant, cfg), ref_v,
                                      "edge N=%d rpw=%d v%d f%d" % (N, rpw, variant, flags))


def test_explicit_geometries_and_flags(pkg, oracle, bundled):
    """Every (vec, strips, group) the launcher accepts gives the same bits, with and
    without the XCD remap, non-temporal stores and 64-bit offsets."""
    from gespmm_amd import _lib

    G = bundled["cora"]
    val = oracle.hash_val(G["nnz"], seed=5)
    for N in (8, 96, 128, 200):
        B = oracle.hash_B(G["K"], N, seed=N)
        ref_v = oracle.spmm(G["rowptr"], G["colind"], val, B, "fma")
        ref_u = oracle.spmm(G["rowptr"], G["colind"], None, B, "golden")
        all_flags = (0, _lib.FLAG_NO_XCD_REMAP, _lib.FLAG_NT_STORE, _lib.FLAG_SC1_STORE, _lib.FLAG_FORCE_IDX64, _lib.FLAG_SEG_STREAM,
                     _lib.FLAG_SHALLOW_UNROLL, _lib.FLAG_BATCH_STREAM,
                     _lib.FLAG_BATCH_STREAM | _lib.FLAG_SHALLOW_UNROLL | _lib.FLAG_FORCE_IDX64,
                     _lib.FLAG_NT_STORE | _lib.FLAG_FORCE_IDX64 | _lib.FLAG_NO_XCD_REMAP | _lib.FLAG_SHALLOW_UNROLL)
        for vec in (1, 2, 4):
            for strips in (1, 2):
                for group in (4, 8, 16, 32, 64):
                    for i, flags in enumerate(all_flags):
                        rpw = (0, 1, 2, 4, 7, 16, 32)[(i + group + vec + strips) % 7]
                        cfg = {"vec": vec, "strips": strips, "group": group, "rows_per_wave": rpw, "flags": flags}
                        what = "N=%d cfg=%r" % (N, cfg)
                        assert_bits_equal(run(pkg, G, B, val, 3, cfg), ref_v, what)
                        assert_bits_equal(run(pkg, G, B, None, 1, cfg), ref_u, what)
                        assert_bits_equal(run(pkg, G, B, val, 0, cfg), ref_v, what + " naive")


def test_parallel_reduction_variant_within_tolerance(pkg, oracle, bundled):
    for g, Ns in (("cora", (1, 2, 3, 4, 7, 8, 12, 16, 20, 32, 41)), ("pubmed", (3, 8))):
        G = bundled[g]
        val = oracle.hash_val(G["nnz"], seed=2)
        for N in Ns:
            B = oracle.hash_B(G["K"], N, seed=N)
            for v in (None, val):
                ref = oracle.spmm(G["rowptr"], G["colind"], v, B, "fma")
                scale = oracle.spmm_abs(G["rowptr"], G["colind"], v, B)
                C = run(pkg, G, B, v, variant=5)
                tol = 1e-4 * np.maximum(np.abs(ref), scale)  # north_star: 1e-4 relative
                assert np.all(np.abs(C.astype(np.float64) - ref) <= tol + 1e-30), (g, N)
    G = edge_case_csr(4)
    B = oracle.hash_B(G["K"], 5, seed=1)
    ref = oracle.spmm(G["rowptr"], G["colind"], None, B, "fma")
    for group in (4, 8, 16, 32, 64):
        C = run(pkg, G, B, None, variant=5, cfg={"group": group})
        assert np.abs(C - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())


def test_reassociation_flag_is_opt_in(pkg, oracle):
    """AUTO stays bit-exact on a dense narrow-N problem; with ALLOW_REASSOCIATION it may
    take the parallel-reduction variant (within tolerance, and equal to variant 5's bits)."""
    from gespmm_amd import _lib

    rng = np.random.RandomState(11)
    M, K = 1500, 2000
    degs = rng.randint(40, 160, size=M)
    rowptr = np.zeros(M + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(degs)
    colind = rng.randint(0, K, size=int(rowptr[-1])).astype(np.int32)
    G = {"M": M, "K": K, "nnz": int(rowptr[-1]), "rowptr": rowptr, "colind": colind}
    val = oracle.hash_val(G["nnz"], seed=4)
    for N in (1, 4, 8, 16):
        B = oracle.hash_B(G["K"], N, seed=N)
        ref = oracle.spmm(G["rowptr"], G["colind"], val, B, "fma")
        scale = oracle.spmm_abs(G["rowptr"], G["colind"], val, B)
        assert_bits_equal(run(pkg, G, B, val, -1), ref, "auto N=%d" % N)
        C = run(pkg, G, B, val, -1, cfg={"flags": _lib.FLAG_ALLOW_REASSOCIATION})
        assert np.all(np.abs(C.astype(np.float64) - ref) <= 1e-4 * np.maximum(np.abs(ref), scale) + 1e-30)
        assert_bits_equal(C, run(pkg, G, B, val, 5), "reassoc == v5, N=%d" % N)


def test_misaligned_and_strided_inputs(pkg, oracle, bundled):
    """A B/C pointer that is only 4- or 8-byte aligned degrades the vector width, never
    the result; non-contiguous inputs are rejected like the reference's asserts."""
    from gespmm_amd import spmm

    G = bundled["citeseer"]
    rp, ci = dev_csr(G)
    N = 64
    B = oracle.hash_B(G["K"], N, seed=8)
    ref = oracle.spmm(G["rowptr"], G["colind"], None, B, "golden")
    for shift in (1, 2, 3):
        flat = torch.zeros(G["K"] * N + 8, dtype=torch.float32, device="cuda")
        Bd = flat[shift:shift + G["K"] * N].view(G["K"], N)
        Bd.copy_(torch.from_numpy(B))
        assert Bd.data_ptr() % 16 != 0
        oflat = torch.empty(G["M"] * N + 8, dtype=torch.float32, device="cuda")
        out = oflat[shift:shift + G["M"] * N].view(G["M"], N)
        spmm.csr_spmm_no_edge_value(rp, ci, Bd, out=out)
        assert_bits_equal(out.cpu().numpy(), ref, "shift %d" % shift)
    with pytest.raises(ValueError):
        spmm.csr_spmm_no_edge_value(rp, ci, torch.from_numpy(B).cuda().t())
    with pytest.raises(TypeError):
        spmm.csr_spmm_no_edge_value(rp.long(), ci, torch.from_numpy(B).cuda())
    with pytest.raises(RuntimeError):
        spmm.csr_spmm_no_edge_value(rp.cpu(), ci, torch.from_numpy(B).cuda())


def test_degenerate_shapes(pkg, oracle):
    from gespmm_amd import spmm

    # M = 0, N = 0, nnz = 0, single row, single column
    rp = torch.zeros(1, dtype=torch.int32, device="cuda")
    ci = torch.zeros(0, dtype=torch.int32, device="cuda")
    assert spmm.csr_spmm_no_edge_value(rp, ci, torch.ones(5, 8, device="cuda")).shape == (0, 8)
    rp = torch.zeros(4, dtype=torch.int32, device="cuda")
    out = spmm.csr_spmm_no_edge_value(rp, ci, torch.ones(5, 8, device="cuda"))
    assert out.shape == (3, 8) and torch.all(out == 0), "rows without non-zeros write 0 (no pre-zeroing needed)"
    assert spmm.csr_spmm_no_edge_value(rp, ci, torch.ones(5, 0, device="cuda")).shape == (3, 0)
    rp = torch.tensor([0, 3], dtype=torch.int32, device="cuda")
    ci = torch.tensor([2, 0, 2], dtype=torch.int32, device="cuda")
    B = torch.arange(3, dtype=torch.float32, device="cuda").view(3, 1) + 1
    v = torch.tensor([0.5, 2.0, -1.0], device="cuda")
    assert spmm.csr_spmm(rp, ci, v, B).item() == 0.5 * 3 + 2.0 * 1 - 3.0
    # output buffer pre-filled with garbage is fully overwritten
    G = edge_case_csr(5)
    Bn = oracle.hash_B(G["K"], 33, seed=1)
    junk = torch.full((G["M"], 33), float("nan"), device="cuda")
    rpd, cid = dev_csr(G)
    spmm.csr_spmm_no_edge_value(rpd, cid, torch.from_numpy(Bn).cuda(), out=junk)
    assert_bits_equal(junk.cpu().numpy(), oracle.spmm(G["rowptr"], G["colind"], None, Bn, "golden"), "overwrite")


def test_special_values_propagate_like_the_reference(pkg, oracle):
    """inf / nan / -0 in B and val: the chain is the same IEEE fma sequence."""
    G = edge_case_csr(6)
    N = 16
    B = oracle.hash_B(G["K"], N, seed=2)
    B[3, :] = np.inf
    B[7, 2] = np.nan
    B[11, :] = -0.0
    val = oracle.hash_val(G["nnz"], seed=4)
    val[::17] = 0.0
    val[5::29] = -0.0
    ref = oracle.spmm(G["rowptr"], G["colind"], val, B, "fma")
    for variant in EXACT_VARIANTS:
        C = run(pkg, G, B, val, variant)
        nan_ref, nan_c = np.isnan(ref), np.isnan(C)
        assert np.array_equal(nan_ref, nan_c)
        assert np.array_equal(bits(C)[~nan_c], bits(ref)[~nan_ref])


def test_max_reducer(pkg, oracle, bundled):
    from gespmm_amd import spmm

    for G in (edge_case_csr(7), bundled["cora"]):
        rp, ci = dev_csr(G)
        for N in (1, 5, 32, 100, 128):
            B = oracle.hash_B(G["K"], N, seed=N)
            for init in (-10000.0, float("-inf")):
                ref = oracle.spmm_max(G["rowptr"], G["colind"], B, init)
                for variant in (-1, 1, 2, 3, 4):
                    C = spmm.csr_spmm_max(rp, ci, torch.from_numpy(B).cuda(), init, variant).cpu().numpy()
                    assert_bits_equal(C, ref, "max N=%d init=%r v%d" % (N, init, variant))


def test_dgl_entry_points(pkg, oracle, bundled):
    """gespmm_dgl_csrmm_{sum,max}_f32 take exactly what the DGL patch's XTopoCsrmm /
    XTopoCsrmmmax receive (no K, no nnz): same bits as the oracle's golden loop."""
    from gespmm_amd import _lib

    for G in (bundled["cora"], edge_case_csr(5), _skewed_csr(2)[0]):
        rp, ci = dev_csr(G)
        for N in (7, 64, 128):
            B = oracle.hash_B(G["K"], N, seed=N + 3)
            Bd = torch.from_numpy(B).cuda()
            out = torch.empty(G["M"], N, dtype=torch.float32, device="cuda")
            st = torch.cuda.current_stream().cuda_stream
            rc = _lib.lib.gespmm_dgl_csrmm_sum_f32(G["M"], N, rp.data_ptr(), ci.data_ptr(), Bd.data_ptr(),
                                                   out.data_ptr(), st)
            assert rc == 0
            torch.cuda.synchronize()
            assert_bits_equal(out.cpu().numpy(), oracle.spmm(G["rowptr"], G["colind"], None, B, "golden"), "dgl sum")
            rc = _lib.lib.gespmm_dgl_csrmm_max_f32(G["M"], N, rp.data_ptr(), ci.data_ptr(), Bd.data_ptr(),
                                                   out.data_ptr(), st)
            assert rc == 0
            torch.cuda.synchronize()
            assert_bits_equal(out.cpu().numpy(), oracle.spmm_max(G["rowptr"], G["colind"], B), "dgl max")
    # the read-back threshold is a process-wide policy: "never" keeps every call asynchronous (strict chains), same bits here
    G = bundled["cora"]
    rp, ci = dev_csr(G)
    B = oracle.hash_B(G["K"], 64, seed=1)
    Bd = torch.from_numpy(B).cuda()
    out = torch.empty(G["M"], 64, dtype=torch.float32, device="cuda")
    for rows in (-1, 0, 1 << 15):
        assert _lib.lib.gespmm_dgl_set_readback_rows(rows) == 0
        assert _lib.lib.gespmm_dgl_csrmm_sum_f32(G["M"], 64, rp.data_ptr(), ci.data_ptr(), Bd.data_ptr(), out.data_ptr(),
                                                 torch.cuda.current_stream().cuda_stream) == 0
        torch.cuda.synchronize()
        assert_bits_equal(out.cpu().numpy(), oracle.spmm(G["rowptr"], G["colind"], None, B, "golden"), "dgl sum, readback %d" % rows)
    # a large dense graph: the entry point reads nnz back and takes the cache-blocked path (K taken as m)
    from gespmm_amd import spmm

    rng = np.random.RandomState(4)
    M, N = 60000, 128
    deg = rng.randint(110, 160, size=M)  # >= 20 entries of a row per 6 MB slab: the cache-blocked path
    rowptr = np.zeros(M + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(deg)
    rp = torch.from_numpy(rowptr).cuda()
    ci = torch.from_numpy(rng.randint(0, M + 5000, size=int(rowptr[-1])).astype(np.int32)).cuda()  # K > m
    Bd = torch.rand(M + 5000, N, device="cuda") - 0.5
    out = torch.empty(M, N, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert _lib.lib.gespmm_dgl_csrmm_sum_f32(M, N, rp.data_ptr(), ci.data_ptr(), Bd.data_ptr(), out.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, spmm.csr_spmm_no_edge_value(rp, ci, Bd, cfg={"flags": _lib.FLAG_NO_SLAB_BLOCKED}))
    assert _lib.lib.gespmm_dgl_csrmm_max_f32(M, N, rp.data_ptr(), ci.data_ptr(), Bd.data_ptr(), out.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, spmm.csr_spmm_max(rp, ci, Bd))


def test_stream_semantics(pkg, oracle, bundled):
    """Launches go to the CURRENT torch stream (the reference uses the legacy default
    stream, spmm_kernel.cu:189,196,203)."""
    from gespmm_amd import spmm

    G = bundled["cora"]
    rp, ci = dev_csr(G)
    B = torch.from_numpy(oracle.hash_B(G["K"], 64, seed=3)).cuda()
    ref = spmm.csr_spmm_no_edge_value(rp, ci, B)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        big = torch.randn(4096, 4096, device="cuda")
        for _ in range(4):
            big = big @ big  # keep the side stream busy ahead of the SpMM
            big = big / big.abs().max()
        x = (big[: G["K"], :64] * 0 + B).contiguous()  # depends on the matmuls: only valid in-stream
        out = spmm.csr_spmm_no_edge_value(rp, ci, x)
    s.synchronize()
    assert torch.equal(out, ref)


def _skewed_csr(seed=0):
    """A few hub rows (2049 .. 70 000 entries, around the 2048-entry split threshold)
    among short and empty rows."""
    rng = np.random.RandomState(seed)
    K = 5000
    degs = rng.randint(0, 12, size=300)
    for r, d in ((0, 2048), (5, 2049), (17, 5000), (18, 70000), (150, 2047), (299, 9999)):
        degs[r] = d
    rowptr = np.zeros(len(degs) + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(degs)
    colind = rng.randint(0, K, size=int(rowptr[-1])).astype(np.int32)
    long_rows = np.nonzero(degs > 2048)[0]
    return {"M": len(degs), "K": K, "nnz": int(rowptr[-1]), "rowptr": rowptr, "colind": colind}, long_rows


def test_long_row_split_is_deterministic_and_within_tolerance(pkg, oracle):
    """Rows above 2048 entries are re-associated (fixed order) when the long-row pass
    runs; every other row stays bit-exact; STRICT_ORDER restores the strict chain."""
    from gespmm_amd import _lib

    G, long_rows = _skewed_csr()
    short = np.setdiff1d(np.arange(G["M"]), long_rows)
    val = oracle.hash_val(G["nnz"], seed=11)
    for N in (3, 32, 128, 200, 512):
        B = oracle.hash_B(G["K"], N, seed=N)
        for v in (val, None):
            ref = oracle.spmm(G["rowptr"], G["colind"], v, B, "fma")
            scale = oracle.spmm_abs(G["rowptr"], G["colind"], v, B)
            split = {"flags": _lib.FLAG_SPLIT_LONG_ROWS}
            for variant in (-1, 1, 2, 3, 4):
                C1 = run(pkg, G, B, v, variant, split)
                C2 = run(pkg, G, B, v, variant, split)
                assert np.array_equal(bits(C1), bits(C2)), "split rows must be reproducible run to run"
                assert np.array_equal(bits(C1[short]), bits(ref[short])), "rows <= 2048 entries stay bit-exact"
                tol = 1e-4 * np.maximum(np.abs(ref[long_rows]), scale[long_rows])  # north_star: 1e-4 relative
                assert np.all(np.abs(C1[long_rows].astype(np.float64) - ref[long_rows]) <= tol + 1e-30), (N, variant)
            strict = {"flags": _lib.FLAG_SPLIT_LONG_ROWS | _lib.FLAG_STRICT_ORDER}
            assert_bits_equal(run(pkg, G, B, v, -1, strict), ref, "STRICT_ORDER N=%d" % N)
            assert_bits_equal(run(pkg, G, B, v, -1, None), ref, "small matrices never split, N=%d" % N)
    # nnz unknown to the caller (the DGL entry point passes -1): the pass
    # cannot size its workspace, so even with the split requested every row keeps the strict chain
    import ctypes

    from gespmm_amd import spmm

    rp, ci = dev_csr(G)
    B = oracle.hash_B(G["K"], 128, seed=5)
    Bd = torch.from_numpy(B).cuda()
    out = torch.empty(G["M"], 128, dtype=torch.float32, device="cuda")
    c = _lib.LaunchCfg()
    c.flags = _lib.FLAG_SPLIT_LONG_ROWS
    rc = _lib.lib.gespmm_csr_spmm_f32_cfg(rp.data_ptr(), ci.data_ptr(), None, Bd.data_ptr(), out.data_ptr(), G["M"],
                                          2**31 - 1, 128, -1, -1, ctypes.byref(c),
                                          torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert_bits_equal(out.cpu().numpy(), oracle.spmm(G["rowptr"], G["colind"], None, B, "golden"), "nnz = -1")
    # many launches back to back: the stream-ordered workspace is recycled, results identical
    val_d = torch.from_numpy(val).cuda()
    first = spmm.csr_spmm(rp, ci, val_d, Bd, cfg=split).clone()
    for _ in range(20):
        again = spmm.csr_spmm(rp, ci, val_d, Bd, cfg=split)
    assert torch.equal(first, again)
    # max reducer: exact under any association
    Bm = oracle.hash_B(G["K"], 64, seed=3)
    assert_bits_equal(spmm.csr_spmm_max(rp, ci, torch.from_numpy(Bm).cuda()).cpu().numpy(),
                      oracle.spmm_max(G["rowptr"], G["colind"], Bm), "max")


def test_hip_graph_capture(pkg, oracle):
    """Captured into a HIP graph nothing may be allocated by the library. Through the bindings
    the scratch of the slab-blocked path and the long-row pass is a framework tensor
    (gespmm_csr_spmm_f32_ws): both paths run inside the graph. Through the plain entry point
    (no workspace) they fall back to the streaming kernel / the strict chain. Either way the
    replay is cheap and correct."""
    import ctypes

    from gespmm_amd import _lib, spmm

    G, long_rows = _skewed_csr(3)
    rp, ci = dev_csr(G)
    val = oracle.hash_val(G["nnz"], seed=2)
    B = oracle.hash_B(G["K"], 128, seed=9)
    ref = oracle.spmm(G["rowptr"], G["colind"], val, B, "fma")
    scale = oracle.spmm_abs(G["rowptr"], G["colind"], val, B)
    vd, Bd = torch.from_numpy(val).cuda(), torch.from_numpy(B).cuda()

    def capture(fn):
        fn()  # eager warm-up
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        return graph

    for flags in (_lib.FLAG_SPLIT_LONG_ROWS, _lib.FLAG_SLAB_BLOCKED):
        cfg = {"flags": flags, "slab_rows": 500}
        out = torch.zeros(G["M"], 128, device="cuda")
        eager = spmm.csr_spmm(rp, ci, vd, Bd, cfg=cfg).clone()
        graph = capture(lambda: spmm.csr_spmm(rp, ci, vd, Bd, cfg=cfg, out=out))
        out.zero_()
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager), "binding: the captured call takes the same path as the eager one"
        got = out.cpu().numpy()
        short = np.setdiff1d(np.arange(G["M"]), long_rows)
        assert np.array_equal(bits(got[short]), bits(ref[short]))
        tol = 1e-4 * np.maximum(np.abs(ref[long_rows]), scale[long_rows])
        assert np.all(np.abs(got[long_rows].astype(np.float64) - ref[long_rows]) <= tol + 1e-30)

        # plain C entry point, no workspace: strict / streaming fallback under capture, bit-exact everywhere
        c = _lib.LaunchCfg()
        c.flags, c.slab_rows = flags, 500
        out2 = torch.zeros(G["M"], 128, device="cuda")

        def raw():
            rc = _lib.lib.gespmm_csr_spmm_f32_cfg(rp.data_ptr(), ci.data_ptr(), vd.data_ptr(), Bd.data_ptr(),
                                                  out2.data_ptr(), G["M"], G["K"], 128, G["nnz"], -1, ctypes.byref(c),
                                                  torch.cuda.current_stream().cuda_stream)
            assert rc == 0

        graph2 = capture(raw)
        out2.zero_()
        graph2.replay()
        torch.cuda.synchronize()
        assert_bits_equal(out2.cpu().numpy(), ref, "plain entry point under capture, flags=%#x" % flags)


def test_slab_blocked_path_is_bit_exact(pkg, oracle, bundled):
    """The cache-blocked kernel for dense graphs consumes every row's entries in CSR
    order whatever the slab size — sorted, unsorted and repeated columns alike."""
    from gespmm_amd import _lib

    cases = [edge_case_csr(8), _skewed_csr(1)[0], bundled["cora"]]
    for G in cases:
        val = oracle.hash_val(G["nnz"], seed=13)
        for N in (1, 3, 32, 64, 100, 128, 512):
            B = oracle.hash_B(G["K"], N, seed=N + 1)
            ref_v = oracle.spmm(G["rowptr"], G["colind"], val, B, "fma")
            ref_u = oracle.spmm(G["rowptr"], G["colind"], None, B, "golden")
            for slab_rows, R in ((1, 1), (7, 3), (64, 8), (1000, 16), (0, 0), (1 << 20, 32)):
                cfg = {"slab_rows": slab_rows, "rows_per_wave": R, "flags": _lib.FLAG_SLAB_BLOCKED}
                for variant in (-1, 1, 3, 4):
                    what = "slab blocked N=%d slab_rows=%d R=%d v%d" % (N, slab_rows, R, variant)
                    assert_bits_equal(run(pkg, G, B, val, variant, cfg), ref_v, what)
                assert_bits_equal(run(pkg, G, B, None, -1, cfg), ref_u, "slab blocked unweighted N=%d" % N)
    # forced 64-bit offsets and explicit geometry
    G = cases[0]
    B = oracle.hash_B(G["K"], 96, seed=2)
    ref = oracle.spmm(G["rowptr"], G["colind"], None, B, "golden")
    for vec, group in ((1, 64), (2, 16), (4, 8), (4, 32)):
        cfg = {"vec": vec, "group": group, "slab_rows": 50,
               "flags": _lib.FLAG_SLAB_BLOCKED | _lib.FLAG_FORCE_IDX64}
        assert_bits_equal(run(pkg, G, B, None, 3, cfg), ref, "slab blocked cfg %r" % cfg)


def test_extension_and_ctypes_bindings_agree(pkg, oracle, bundled):
    """Default calls go through the pybind11 extension, calls with tuning knobs through
    ctypes: same C ABI, same bits, same exception types."""
    from gespmm_amd import _ext, spmm

    G = bundled["citeseer"]
    rp, ci = dev_csr(G)
    B = torch.from_numpy(oracle.hash_B(G["K"], 48, seed=4)).cuda()
    val = torch.from_numpy(oracle.hash_val(G["nnz"], seed=6)).cuda()
    a = spmm.csr_spmm(rp, ci, val, B)                       # extension when built
    b = spmm.csr_spmm(rp, ci, val, B, cfg={"flags": 0})     # ctypes
    assert torch.equal(a, b)
    ref = oracle.spmm(G["rowptr"], G["colind"], val.cpu().numpy(), B.cpu().numpy(), "fma")
    assert_bits_equal(a.cpu().numpy(), ref, "extension path")
    if _ext.ext is None:
        pytest.skip("extension not built")
    for bad, exc in ((lambda: _ext.ext.csr_spmm_no_edge_value(rp, ci, B.t(), -1), ValueError),
                     (lambda: _ext.ext.csr_spmm_no_edge_value(rp.long(), ci, B, -1), TypeError),
                     (lambda: _ext.ext.csr_spmm_no_edge_value(rp.cpu(), ci, B, -1), RuntimeError),
                     (lambda: _ext.ext.csr_spmm(rp, ci, val[:-1], B, -1), ValueError),
                     (lambda: _ext.ext.csr_spmm_no_edge_value(rp, ci, B, 17), RuntimeError)):
        with pytest.raises(exc):
            bad()


def test_very_short_rows_take_the_segmented_kernel_and_keep_the_bits(pkg, oracle):
    """Mean degree <= 3 on a big matrix (road-network-like): AUTO picks the segmented-stream kernel (select.cpp);
    the bits are the oracle's, empty rows included."""
    import ctypes

    from gespmm_amd import _lib

    def describe(M, K, N, nnz):
        buf = ctypes.create_string_buffer(256)
        assert _lib.lib.gespmm_describe_launch(M, K, N, nnz, -1, None, buf, 256) > 0
        return buf.value.decode()

    rng = np.random.default_rng(11)
    M = K = 70001
    deg = rng.integers(0, 5, size=M)  # 0..4 entries, mean 2
    rowptr = np.zeros(M + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(deg)
    nnz = int(rowptr[-1])
    colind = rng.integers(0, K, size=nnz).astype(np.int32)
    for r in rng.integers(0, M, size=2000):  # sorted or not must not matter; sort a few rows only
        colind[rowptr[r]:rowptr[r + 1]].sort()
    G = {"rowptr": rowptr, "colind": colind, "M": M, "K": K, "nnz": nnz}
    val = oracle.hash_val(nnz, seed=3)
    for N in (128, 100, 256):
        what = describe(M, K, N, nnz)
        assert "segmented-stream" in what, what
        B = oracle.hash_B(K, N, seed=2)
        assert_bits_equal(run(pkg, G, B, val), oracle.spmm(rowptr, colind, val, B, "fma"), "short rows N=%d valued" % N)
        assert_bits_equal(run(pkg, G, B, None), oracle.spmm(rowptr, colind, None, B, "golden"), "short rows N=%d" % N)


def test_l2_resident_b_with_large_c_keeps_the_bits(pkg, oracle):
    """Many rows over a small B (K = 2048: sampled-neighbour / bipartite shapes), with and without system-scope C stores
    (GESPMM_FLAG_SC1_STORE): a store flavour never changes a bit."""
    rng = np.random.default_rng(3)
    M, K = 40000, 2048
    deg = rng.integers(0, 12, size=M)
    rowptr = np.zeros(M + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(deg)
    nnz = int(rowptr[-1])
    colind = rng.integers(0, K, size=nnz).astype(np.int32)
    G = {"rowptr": rowptr, "colind": colind, "M": M, "K": K, "nnz": nnz}
    val = oracle.hash_val(nnz, seed=5)
    for N in (64, 128):
        B = oracle.hash_B(K, N, seed=N)
        ref = oracle.spmm(rowptr, colind, val, B, "fma")
        assert_bits_equal(run(pkg, G, B, val), ref, "small B, N=%d" % N)
        assert_bits_equal(run(pkg, G, B, val, cfg={"flags": 0x20}), ref, "small B, batch kernel, N=%d" % N)
        assert_bits_equal(run(pkg, G, B, val, cfg={"flags": 0x8000}), ref, "small B, sc1 stores, N=%d" % N)
        assert_bits_equal(run(pkg, G, B, val, cfg={"flags": 0x8020}), ref, "small B, batch kernel, sc1 stores, N=%d" % N)
