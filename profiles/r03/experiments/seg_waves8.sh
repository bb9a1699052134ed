#!/bin/bash
# EXPERIMENT: amdgpu_waves_per_eu(8) on the scalar (V = 1) segmented-stream instantiations behind a plan (65 -> 62 VGPRs at W = 16).
cd $GRAFT_REPO_ROOT
run() { python - <<PY
import sys, statistics, torch
sys.path.insert(0, ".")
import gespmm_amd
from gespmm_amd import graphs, spmm
for name in ("products-sbm", "com-amazon-sbm"):
    g = graphs.synthetic_graph(name, seed=42, device="cuda")
    val = torch.rand(g["nnz"], device="cuda") - 0.5
    for N in (16, 32):
        B = torch.rand(g["K"], N, device="cuda") - 0.5
        C = torch.empty(g["M"], N, device="cuda")
        plan = spmm.SpmmPlan(g["rowptr"], g["colind"], g["K"], N, values=val, kernel="seg-stream")
        for _ in range(3): spmm.csr_spmm(g["rowptr"], g["colind"], val, B, out=C, plan=plan)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in ev:
            a.record(); spmm.csr_spmm(g["rowptr"], g["colind"], val, B, out=C, plan=plan); b.record()
        torch.cuda.synchronize()
        print("%-15s N=%3d seg-stream %8.1f us" % (name, N, statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev)), flush=True)
PY
}
echo "== with amdgpu_waves_per_eu(8) on V=1 kernels"; 
HIPCC=/opt/rocm/bin/hipcc
cp gespmm_amd/lib/libgespmm.so /tmp/libgespmm.orig.so
$HIPCC -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -w -DGESPMM_EXP_WAVES8 -c gespmm_amd/csrc/spmm_stream_plan.hip -o /tmp/ssp_w8.o && \
$HIPCC --offload-arch=gfx950 -shared -fPIC -o gespmm_amd/lib/libgespmm.so $(ls gespmm_amd/lib/obj/*.o | grep -v spmm_stream_plan.o) /tmp/ssp_w8.o
run
cp /tmp/libgespmm.orig.so gespmm_amd/lib/libgespmm.so
echo "== shipped"; run
