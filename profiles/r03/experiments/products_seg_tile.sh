#!/bin/bash
# EXPERIMENT: segmented-stream kernel with 64-entry tiles behind a plan (-DGESPMM_EXP_SEG_TILE64 on spmm_stream_plan.hip only).
cd $GRAFT_REPO_ROOT
run() { python - <<PY
import sys, statistics, torch
sys.path.insert(0, ".")
import gespmm_amd
from gespmm_amd import graphs, spmm
g = graphs.synthetic_graph("products-sbm", seed=42, device="cuda")
val = torch.rand(g["nnz"], device="cuda") - 0.5
for N in (16, 32, 128):
    B = torch.rand(g["K"], N, device="cuda") - 0.5
    C = torch.empty(g["M"], N, device="cuda")
    for kern in ("auto", "seg-stream", "stream"):
        plan = spmm.SpmmPlan(g["rowptr"], g["colind"], g["K"], N, values=val, kernel=kern)
        for _ in range(3): spmm.csr_spmm(g["rowptr"], g["colind"], val, B, out=C, plan=plan)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, b in ev:
            a.record(); spmm.csr_spmm(g["rowptr"], g["colind"], val, B, out=C, plan=plan); b.record()
        torch.cuda.synchronize()
        print("N=%3d kernel=%-10s %8.1f us" % (N, kern, statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev)), flush=True)
PY
}
echo "== shipped (32-entry tiles)"; run
HIPCC=/opt/rocm/bin/hipcc
$HIPCC -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -DGESPMM_EXP_SEG_TILE64 -c gespmm_amd/csrc/spmm_stream_plan.hip -o /tmp/ssp64.o && \
$HIPCC --offload-arch=gfx950 -shared -fPIC -o gespmm_amd/lib/libgespmm.so $(ls gespmm_amd/lib/obj/*.o | grep -v spmm_stream_plan.o) /tmp/ssp64.o
echo "== 64-entry tiles behind a plan"; run
