"""Clustering knob sweep: plan time, modelled hits and kernel time per setting.
python scripts/plan_knobs.py graph N"""
import os, subprocess, sys
graph = sys.argv[1] if len(sys.argv) > 1 else "com-amazon-sbm"
N = sys.argv[2] if len(sys.argv) > 2 else "128"
child = r'''
import sys, time, statistics, torch
sys.path.insert(0, ".")
import gespmm_amd
from gespmm_amd import graphs, spmm
g = graphs.synthetic_graph(sys.argv[1], seed=42, device="cuda"); N = int(sys.argv[2])
val = torch.rand(g["nnz"], device="cuda") - 0.5
B = torch.rand(g["K"], N, device="cuda") - 0.5
C = torch.empty(g["M"], N, device="cuda")
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    plan = spmm.SpmmPlan(g["rowptr"], g["colind"], g["K"], N, values=val)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
for _ in range(5): spmm.csr_spmm(g["rowptr"], g["colind"], val, B, out=C, plan=plan)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
for a, b in ev:
    a.record(); spmm.csr_spmm(g["rowptr"], g["colind"], val, B, out=C, plan=plan); b.record()
torch.cuda.synchronize()
us = statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev)
d = plan.describe()
print("plan_ms %.2f (first %.1f) kernel_us %.1f | %s" % (min(ts), ts[0], us, d[:150]))
'''
settings = [{}, {"STOP": "90"}, {"STOP": "80"}, {"STOP": "70"}, {"LEVELS": "3"}, {"LEVELS": "2"}, {"SWEEPS": "3"}, {"SWEEPS": "3", "STOP": "80"},
            {"SWEEPS": "2", "STOP": "80"}, {"SWEEPS": "3", "LEVELS": "3"}, {"SWEEPS": "2", "LEVELS": "2"}]
for s in settings:
    env = dict(os.environ)
    for k, v in s.items():
        env["GESPMM_CLUSTER_" + k] = v
    out = subprocess.run([sys.executable, "-c", child, graph, N], env=env, capture_output=True, text=True)
    print("%-36s %s" % (s, (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1]), flush=True)
