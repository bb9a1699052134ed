"""EXPERIMENT: cluster-blocked LDS walk (spmm_hotrows.hip, slab_walk_kernel) on dense clustered graphs, N = 128.
    python slabwalk_time.py [reddit-sbm|products-sbm|com-amazon-sbm] [R ...]"""
import ctypes, os, statistics, sys
import torch
sys.path.insert(0, ".")
import gespmm_amd
from gespmm_amd import graphs, spmm

HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "_build", "libhotrows.so"))
lib.slabwalk_spmm.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 7 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
name = sys.argv[1] if len(sys.argv) > 1 else "reddit-sbm"
Rs = [int(x) for x in sys.argv[2:]] or [96, 64, 128]
N, H = 128, 128
if name == "reddit-sbm":
    M, nnz0 = graphs.SPECS["reddit-like"][:2]
    rp, ci, _ = graphs.community_csr(M, nnz0, 290, 16, 330.0, 0.6, 1.5, 1.55, 42, "cuda")
    K = M
else:
    g = graphs.synthetic_graph(name, seed=42, device="cuda")
    M, K, rp, ci = g["M"], g["K"], g["rowptr"], g["colind"]
nnz = int(ci.numel())
val = torch.rand(nnz, device="cuda") - 0.5
B = torch.rand(K, N, device="cuda") - 0.5
C = torch.empty(M, N, device="cuda")
plan = spmm.SpmmPlan(rp, ci, K, N, values=val)
print(name, "M", M, "nnz", nnz, "|", plan.describe()[:200])


def timed(fn, reps=7):
    for _ in range(2): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev)


t_plan = timed(lambda: spmm.csr_spmm(rp, ci, val, B, out=C, plan=plan))
want = C.clone()
print("library, plan path: %.1f us" % t_plan)
perm = plan.order().cuda().to(torch.int64) if plan.clustered else torch.arange(M, device="cuda")
deg = (rp[1:] - rp[:-1]).to(torch.int64)
lens = deg[perm]
rp_p = torch.zeros(M + 1, dtype=torch.int64, device="cuda")
rp_p[1:] = torch.cumsum(lens, 0)
src = torch.repeat_interleave(rp[:-1].to(torch.int64)[perm] - rp_p[:-1], lens) + torch.arange(nnz, device="cuda")
ci_p = ci[src].to(torch.int64)
val_p = val[src]
row_p = torch.repeat_interleave(torch.arange(M, device="cuda"), lens)
del src
stream = torch.cuda.current_stream().cuda_stream
perm32 = perm.to(torch.int32)
for R in Rs:
    nblk = (M + R - 1) // R
    blk = row_p // R
    key = blk * K + ci_p
    u, inv, ucnt = torch.unique(key, return_inverse=True, return_counts=True)  # ascending: by block, then by column
    ublk = u // K
    mixed = 0 if os.environ.get("ALLSTAGED") else 1
    hot_u = (ucnt >= 2) if mixed else torch.ones_like(ucnt, dtype=torch.bool)
    # the block's HOT columns, ascending, cut into slabs of H; an entry belongs to the slab whose column range holds its column
    hot_rank_u = torch.cumsum(hot_u.to(torch.int64), 0) - hot_u.to(torch.int64)        # hot columns before u (global)
    first_u = torch.zeros(nblk + 1, dtype=torch.int64, device="cuda")
    first_u[1:] = torch.cumsum(torch.bincount(ublk, minlength=nblk), 0)
    hot_before_blk = torch.cat([hot_rank_u, hot_rank_u[-1:] + hot_u[-1:].to(torch.int64)])[first_u]  # hot columns before each block
    d_u = hot_rank_u - hot_before_blk[ublk]                                              # rank among the block's hot columns (cold: hot ones below it)
    nhot_blk = hot_before_blk[1:] - hot_before_blk[:-1]
    slab_u = torch.minimum(d_u // H, torch.clamp((nhot_blk[ublk] - 1) // H, min=0))       # cold columns above the last hot one: last slab
    slot_u = d_u % H
    slab, hot_e = slab_u[inv], hot_u[inv]
    code = torch.where(hot_e, slot_u[inv] | (1 << 31), ci_p)
    code32 = (code & 0xFFFFFFFF)
    code32 = torch.where(code32 >= (1 << 31), code32 - (1 << 32), code32).to(torch.int32)
    NS = int(torch.clamp((nhot_blk + H - 1) // H, min=1).max())
    assert bool((slab[1:] >= slab[:-1])[row_p[1:] == row_p[:-1]].all()), "rows must have ascending columns"
    cnt = torch.bincount(row_p * (NS + 1) + slab + 1, minlength=M * (NS + 1)).view(M, NS + 1)
    split = (rp_p[:-1, None] + torch.cumsum(cnt, 1)).to(torch.int32).contiguous()
    ev = torch.cat([torch.stack([code32, val_p.view(torch.int32)], 1), torch.zeros(64, 2, dtype=torch.int32, device="cuda")]).contiguous()
    hu = torch.nonzero(hot_u).squeeze(1)
    ucols = torch.cat([(u[hu] % K).to(torch.int32), torch.zeros(H, dtype=torch.int32, device="cuda")])
    uoff = hot_before_blk
    uoff32 = uoff.to(torch.int32)
    staged_share = float(hot_e.float().mean())
    fn = lambda: lib.slabwalk_spmm(R, split.data_ptr(), ev.data_ptr(), perm32.data_ptr(), ucols.data_ptr(), uoff32.data_ptr(), B.data_ptr(),
                                   C.data_ptr(), nblk, NS, M, mixed, stream)
    C.zero_()
    rc = fn(); torch.cuda.synchronize()
    same = bool(torch.equal(C.view(torch.int32), want.view(torch.int32)))
    if not same:
        bad = (C.view(torch.int32) != want.view(torch.int32)).any(1)
        print("   %d rows differ; max |diff| %.3g" % (int(bad.sum()), float((C - want).abs().max())))
    t = timed(fn)
    print("R=%3d: %d blocks, staged columns per block %.0f (%.1f %% of the entries), slabs per block <= %d (mean %.1f): %8.1f us  x%.2f vs the library  bits=%s rc=%d"
          % (R, nblk, float(nhot_blk.float().mean()), 100 * staged_share, NS, float(((nhot_blk + H - 1) // H).float().mean()), t, t_plan / t, same, rc), flush=True)
    del key, u, inv, ublk, slab, cnt, split, ev
