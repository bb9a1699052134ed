"""Probe of tests/test_gpu_graph_capture.py::test_plain_calls_and_a_two_layer_propagation_replay_from_a_graph (fails about every second time
when it is the first thing a process does): which rows of Y differ, over several replays, with and without the plan in layer 1."""
import sys
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import gespmm_amd
from gespmm_amd import _lib, graphs, spmm
import oracle_py as oracle

mode = sys.argv[1] if len(sys.argv) > 1 else "test"
coo = oracle.read_mtx("tests/golden/cora.mtx")
rowptr, colind, _ = oracle.coo_to_csr(coo["nrows"], coo["row"], coo["col"])
M, K, nnz = coo["nrows"], coo["ncols"], coo["nnz"]
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
rp, ci = dev(rowptr), dev(colind)
val_h = oracle.hash_val(nnz, seed=5)
val = dev(val_h)
X = dev(oracle.hash_B(K, 128, seed=1))
H = torch.empty((M, 128), device="cuda")
Y = torch.empty((M, 7), device="cuda")
Hs = torch.empty((M, 7), device="cuda")
plan_h = spmm.SpmmPlan(rp, ci, K, 128, values=val)
cfg = {"flags": _lib.FLAG_SPLIT_LONG_ROWS} if mode != "nocfg" else None
print(plan_h.describe()[:200])

def forward():
    spmm.csr_spmm(rp, ci, val, X, out=H, plan=plan_h)
    if mode == "prealloc":
        Hs.copy_(H[:, :7])
        spmm.csr_spmm(rp, ci, val, Hs, out=Y, cfg=cfg)
    else:
        spmm.csr_spmm(rp, ci, val, H[:, :7].contiguous(), out=Y, cfg=cfg)
    return Y

s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): forward()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    forward()
bad_total = 0
for seed in range(2, 10):
    X_h = oracle.hash_B(K, 128, seed=seed)
    X.copy_(dev(X_h)); H.zero_(); Y.zero_()
    graph.replay(); torch.cuda.synchronize()
    H_ref = oracle.spmm(rowptr, colind, val_h, X_h, "fma")
    Y_ref = oracle.spmm(rowptr, colind, val_h, np.ascontiguousarray(H_ref[:, :7]), "fma")
    hb = np.nonzero((H.cpu().numpy().view(np.uint32) != H_ref.view(np.uint32)).any(axis=1))[0]
    yb = np.nonzero((Y.cpu().numpy().view(np.uint32) != Y_ref.view(np.uint32)).any(axis=1))[0]
    bad_total += len(hb) + len(yb)
    print("mode=%s seed %d: H rows differing %s, Y rows differing %s%s" % (mode, seed, hb[:8].tolist(), yb[:8].tolist(),
          (" Y[bad]=%s deg=%s" % (Y[int(yb[0])].cpu().numpy().tolist()[:3], int(rowptr[yb[0] + 1] - rowptr[yb[0]]))) if len(yb) else ""), flush=True)
print("mode=%s: %d differing rows in total" % (mode, bad_total))
