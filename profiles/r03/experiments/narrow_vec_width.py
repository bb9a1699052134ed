"""Planned SpMM at narrow widths: AUTO (V = 1 for N <= 64) against the explicit vector variants.  python narrow_vec_width.py [graph]"""
import statistics, sys, torch
sys.path.insert(0, ".")
import gespmm_amd
from gespmm_amd import graphs, spmm
name = sys.argv[1] if len(sys.argv) > 1 else "products-sbm"
g = graphs.synthetic_graph(name, seed=42, device="cuda")
val = torch.rand(g["nnz"], device="cuda") - 0.5
for N in (16, 32, 64):
    B = torch.rand(g["K"], N, device="cuda") - 0.5
    C = torch.empty(g["M"], N, device="cuda")
    ref = None
    for variant in (-1, 1, 2, 3):
        for kern in ("auto", "stream", "seg-stream"):
            try:
                plan = spmm.SpmmPlan(g["rowptr"], g["colind"], g["K"], N, values=val, variant=variant, kernel=kern)
                for _ in range(3): spmm.csr_spmm(g["rowptr"], g["colind"], val, B, out=C, plan=plan, variant=variant)
            except Exception as ex:
                print("N=%d variant=%d kernel=%s: %s" % (N, variant, kern, str(ex)[:60])); continue
            if ref is None: ref = C.clone()
            same = bool(torch.equal(ref.view(torch.int32), C.view(torch.int32)))
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
            for a, b in ev:
                a.record(); spmm.csr_spmm(g["rowptr"], g["colind"], val, B, out=C, plan=plan, variant=variant); b.record()
            torch.cuda.synchronize()
            print("N=%2d variant=%2d kernel=%-10s %8.1f us  bits=%s | %s" % (N, variant, kern, statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev), same, plan.describe().split("|")[-1][:70]), flush=True)
