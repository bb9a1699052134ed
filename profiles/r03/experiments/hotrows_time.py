"""EXPERIMENT: B rows used more than once by a block of clustered rows staged in LDS (spmm_hotrows.hip) against the
library's plan path. N = 128.   python hotrows_time.py [graph] [R,H,WAVES ...]"""
import ctypes, os, statistics, sys
import torch
sys.path.insert(0, ".")
import gespmm_amd
from gespmm_amd import graphs, spmm

HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "_build", "libhotrows.so"))
lib.hotrows_spmm.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 9 + [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]

name = sys.argv[1] if len(sys.argv) > 1 else "products-sbm"
cfgs = [tuple(int(x) for x in a.split(",")) for a in sys.argv[2:]] or [(64, 128, 8), (128, 128, 8), (64, 64, 8), (128, 128, 16)]
N = 128
g = graphs.synthetic_graph(name, seed=42, device="cuda")
M, K, nnz = g["M"], g["K"], g["nnz"]
rp, ci = g["rowptr"], g["colind"]
val = torch.rand(nnz, device="cuda") - 0.5
B = torch.rand(K, N, device="cuda") - 0.5
C = torch.empty(M, N, device="cuda")
plan = spmm.SpmmPlan(rp, ci, K, N, values=val)
print(name, "M", M, "nnz", nnz, "|", plan.describe())


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
if os.environ.get("TRANSPOSE"):
    # rows of every 128-row block dealt round-robin to the 16 wavefronts: at any moment the block works on CONSECUTIVE rows
    Tw = int(os.environ["TRANSPOSE"])   # wavefronts
    Rb = 128
    full = (M // Rb) * Rb
    idx = torch.arange(full, device="cuda").view(-1, Rb // Tw, Tw).transpose(1, 2).reshape(-1)
    perm = torch.cat([perm[idx], perm[full:]])
deg = (rp[1:] - rp[:-1]).to(torch.int64)
lens = deg[perm]
rp_p = torch.zeros(M + 1, dtype=torch.int64, device="cuda")
rp_p[1:] = torch.cumsum(lens, 0)
src = torch.repeat_interleave(rp[:-1].to(torch.int64)[perm] - rp_p[:-1], lens) + torch.arange(nnz, device="cuda")
ci_p = ci[src].to(torch.int64)
val_p = torch.cat([val[src], torch.zeros(64, device="cuda")])
row_p = torch.repeat_interleave(torch.arange(M, device="cuda"), lens)
del src
stream = torch.cuda.current_stream().cuda_stream

for R, H, WAVES in cfgs:
    nblk = (M + R - 1) // R
    blk = row_p // R
    key = blk * K + ci_p
    u, inv, cnt = torch.unique(key, return_inverse=True, return_counts=True)
    ublk = u // K
    # rank of every (block, column) inside its block by descending use count
    o = torch.argsort(ublk * (1 << 20) + ((1 << 20) - 1 - cnt.clamp(max=(1 << 20) - 1)))
    first = torch.zeros(nblk + 1, dtype=torch.int64, device="cuda")
    first[1:] = torch.cumsum(torch.bincount(ublk, minlength=nblk), 0)
    rank = torch.empty_like(o)
    rank[o] = torch.arange(o.numel(), device="cuda") - first[ublk[o]]
    hot_u = (rank < H) & (cnt >= 2)
    code = torch.where(hot_u[inv], rank[inv] | (1 << 31), ci_p)
    code32 = (code & 0xFFFFFFFF).to(torch.int64)
    code32 = torch.cat([torch.where(code32 >= (1 << 31), code32 - (1 << 32), code32).to(torch.int32), torch.zeros(64, dtype=torch.int32, device="cuda")])
    hot_cols = torch.zeros(nblk * H, dtype=torch.int32, device="cuda")
    hu = torch.nonzero(hot_u).squeeze(1)
    hot_cols[ublk[hu] * H + rank[hu]] = (u[hu] % K).to(torch.int32)
    nhot = torch.bincount(ublk[hu], minlength=nblk).to(torch.int32)
    frac_hot = float(hot_u[inv].float().mean())
    saved = (float(hot_u[inv].sum()) - float(hot_u.sum())) / nnz
    # tasks: the block's rows cut into WAVES parts of about equal entry counts
    b0 = torch.arange(nblk, device="cuda") * R
    b1 = torch.clamp(b0 + R, max=M)
    e0, e1 = rp_p[b0], rp_p[b1]
    bounds = [b0]
    for w in range(1, WAVES):
        target = e0 + (e1 - e0) * w // WAVES
        r = torch.searchsorted(rp_p, target, right=False)
        r = torch.minimum(torch.maximum(r, b0), b1)
        bounds.append(torch.maximum(r, bounds[-1]))
    bounds.append(b1)
    bnd = torch.stack(bounds, 1)  # nblk x (WAVES+1)
    tasks = torch.stack([bnd[:, :-1], bnd[:, 1:] - bnd[:, :-1], rp_p[bnd[:, :-1]], rp_p[bnd[:, 1:]]], 2).to(torch.int32).contiguous()
    assert int((bnd[:, 1:] - bnd[:, :-1]).max()) <= 128
    rp32 = rp_p.to(torch.int32)
    perm32 = perm.to(torch.int32)
    del key, u, inv, cnt, ublk, o, rank, code
    modes = (7, 10, 11, 20, 21) if os.environ.get('FLOOR') else ((10, 12, 13, 14) if os.environ.get('DEBUG') else (7, 10, 11))
    if os.environ.get('MODES'): modes = tuple(int(x) for x in os.environ['MODES'].split(','))
    for mode in modes:
        C.zero_()
        codes = code32
        allstaged = mode >= 16
        if mode >= 16:  # floor of everything but the memory gathers: EVERY entry reads a staged row
            allhot = (((ci_p % torch.clamp(nhot[blk].to(torch.int64), min=1)) | (1 << 31)) & 0xFFFFFFFF)
            allhot = torch.cat([(allhot - (1 << 32)).to(torch.int32), torch.zeros(64, dtype=torch.int32, device="cuda")])
            codes, mode = allhot, mode - 10
        if mode in (10, 11, 12, 13, 14):
            codes = torch.stack([codes, val_p.view(torch.int32)], 1).contiguous()
        fn = lambda: lib.hotrows_spmm(H, WAVES, mode, rp32.data_ptr(), codes.data_ptr() if mode in (0, 1, 3, 6, 7, 10, 11, 12, 13, 14) else ci_p32.data_ptr(),
                                      val_p.data_ptr(), perm32.data_ptr(), tasks.data_ptr(), hot_cols.data_ptr(),
                                      nhot.data_ptr(), B.data_ptr(), C.data_ptr(), nblk, K * N * 4, stream)
        if mode in (2, 4):
            ci_p32 = torch.cat([ci_p.to(torch.int32), torch.zeros(64, dtype=torch.int32, device="cuda")])
        rc = fn()
        torch.cuda.synchronize()
        if rc != 0:
            print("R=%d H=%d waves=%d mode=%d: launch failed rc=%d" % (R, H, WAVES, mode, rc)); continue
        same = bool(torch.equal(C.view(torch.int32), want.view(torch.int32)))
        if not same and not allstaged and os.environ.get("DEBUG"):
            badrows = torch.nonzero((C.view(torch.int32) != want.view(torch.int32)).any(1)).squeeze(1)
            inv_perm = torch.empty_like(perm); inv_perm[perm] = torch.arange(M, device="cuda")
            pos = inv_perm[badrows]
            o2 = torch.argsort(pos)[:12]
            print("   %d rows differ of %d; max |diff| %.3g" % (badrows.numel(), M, float((C - want).abs().max())))
            for i in o2.tolist()[:3]:
                r, pp = int(badrows[i]), int(pos[i])
                tix = int(torch.searchsorted(tasks.view(-1, 4)[:, 0].to(torch.int64).contiguous(), torch.tensor([pp], device="cuda"), right=True)) - 1
                tk = tasks.view(-1, 4)[tix].tolist()
                print("   row %d at position %d (degree %d, entries %d..%d) task %d = %s | got %s want %s" % (r, pp, int(lens[pp]), int(rp_p[pp]), int(rp_p[pp + 1]), tix, tk, C[r, :3].tolist(), want[r, :3].tolist()))
        t = timed(fn)
        print("R=%3d H=%3d waves=%2d mode=%d (%s): %8.1f us  x%.2f vs plan  bits=%s | staged entries %.1f%%, L2 requests saved %.1f%%, mean staged rows %.0f"
              % (R, H, WAVES, mode, ("flat", "lds|global", "nothing staged", "buffer|lds", "scalar-stream U=8", "scalar-stream U=16", "scalar-stream + staged rows U=8", "scalar-stream + staged rows U=16", "", "", "lean scalar + staged U=8", "lean scalar + staged U=16", "lean, C++ fma", "lean, lds base added", "lean, EXEC-masked paths instead of branches")[mode] + (" ALL entries staged (floor)" if allstaged else ""), t, t_plan / t, same, 100 * frac_hot, 100 * saved,
                 float(nhot.float().mean())), flush=True)
