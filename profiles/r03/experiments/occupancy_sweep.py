"""EXPERIMENT: the scalar-stream kernel of spmm_hotrows.hip with the number of resident workgroups per CU limited (through
an unused LDS allocation): is the planned products-shaped product bound by what is in flight (L2 working set) or by
latency hiding?   python occupancy_sweep.py [graph] [task_entries ...]"""
import ctypes, os, statistics, sys
import torch
sys.path.insert(0, ".")
import gespmm_amd
from gespmm_amd import graphs, spmm

HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "_build", "libhotrows.so"))
lib.hotrows_spmm.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 9 + [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
name = sys.argv[1] if len(sys.argv) > 1 else "products-sbm"
budgets = [int(x) for x in sys.argv[2:]] or [256, 512, 1024]
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
deg = (rp[1:] - rp[:-1]).to(torch.int64)
lens = deg[perm]
rp_p = torch.zeros(M + 1, dtype=torch.int64, device="cuda")
rp_p[1:] = torch.cumsum(lens, 0)
src = torch.repeat_interleave(rp[:-1].to(torch.int64)[perm] - rp_p[:-1], lens) + torch.arange(nnz, device="cuda")
ci_p = torch.cat([ci[src], torch.zeros(64, dtype=torch.int32, device="cuda")])
val_p = torch.cat([val[src], torch.zeros(64, device="cuda")])
del src
ci_res = (ci_p % 4096).to(torch.int32)
rp32, perm32 = rp_p.to(torch.int32), perm.to(torch.int32)
stream = torch.cuda.current_stream().cuda_stream
dummy = torch.zeros(16, dtype=torch.int32, device="cuda")
for budget in budgets:
    # tasks: consecutive rows, cut where the running entry count passes a multiple of `budget`
    tid = rp_p[:-1] // budget
    first = torch.ones(M, dtype=torch.bool, device="cuda")
    first[1:] = tid[1:] != tid[:-1]
    starts = torch.nonzero(first).squeeze(1)
    ends = torch.cat([starts[1:], torch.tensor([M], device="cuda")])
    tasks = torch.stack([starts, ends - starts, rp_p[starts], rp_p[ends]], 1).to(torch.int32).contiguous()
    nt = tasks.shape[0]
    for mode, colarr, what in ((4, ci_p, "64 lanes x dwordx2"), (8, ci_p, "32 lanes x dwordx4"), (9, ci_p, "32 lanes x dwordx4 U=16"),
                               (4, ci_res, "64 lanes x dwordx2, B rows from a 2 MB table"), (8, ci_res, "32 lanes x dwordx4, B rows from a 2 MB table"),
                               (9, ci_res, "32 lanes x dwordx4 U=16, B rows from a 2 MB table")):
        if os.environ.get("PMC_ONLY") and (mode != 4 or colarr is not ci_p): continue
        for wgs in ((0, 8, 6, 4, 2) if os.environ.get("PMC_ONLY") else (0,)):
            fn = lambda: lib.hotrows_spmm(wgs, 1, mode, rp32.data_ptr(), colarr.data_ptr(), val_p.data_ptr(), perm32.data_ptr(),
                                          tasks.data_ptr(), dummy.data_ptr(), dummy.data_ptr(), B.data_ptr(), C.data_ptr(), nt,
                                          K * N * 4, stream)
            C.zero_()
            rc = fn(); torch.cuda.synchronize()
            same = bool(torch.equal(C.view(torch.int32), want.view(torch.int32)))
            t = timed(fn)
            print("task_entries~%4d (%d tasks) %s: %8.1f us  x%.2f vs plan  bits=%s  (%.1f TB/s of gathers)" %
                  (budget, nt, what, t, t_plan / t, same, nnz * 512 / t / 1e6), flush=True)
