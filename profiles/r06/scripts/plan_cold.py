#!/usr/bin/env python3
"""Round 6: what the FIRST plan of a process costs, with and without gespmm_init, and the reference's protocol end to end (one process per
matrix, 200 launches: run_test.sh:5-11) through the driver — default method 2, --method -1, --plan (gespmm_init + analysis inside the time).
    python profiles/r06/scripts/plan_cold.py            (spawns fresh processes of itself: --child cold|warm)"""
import ctypes
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)


def child(mode, name):
    import torch

    from gespmm_amd import _lib, graphs

    g = graphs.synthetic_graph(name, seed=42, device="cuda")
    rp, ci, K, M, nnz = g["rowptr"], g["colind"], g["K"], g["M"], g["nnz"]
    val = torch.rand(nnz, device="cuda") - 0.5
    torch.cuda.synchronize()
    t_init = 0.0
    if mode == "warm":
        t0 = time.perf_counter()
        _lib.init(M, nnz)
        torch.cuda.synchronize()
        t_init = (time.perf_counter() - t0) * 1e3
    ts = []
    desc = ""
    for i in range(4):
        h = ctypes.c_void_p()
        # forced clustering for the FIRST creation (a cold AUTO plan now declines: the cost rule adds the cold cost), AUTO afterwards
        opt = _lib.PlanOptions(_lib.PLAN_REORDER if i == 0 else _lib.PLAN_REORDER_AUTO, 0, 0, 0, 0, 0, 0, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = _lib.lib.gespmm_plan_create_v2(ctypes.byref(h), ctypes.c_void_p(rp.data_ptr()), ctypes.c_void_p(ci.data_ptr()),
                                            ctypes.c_void_p(val.data_ptr()), M, K, nnz, 128, -1, ctypes.byref(opt), ctypes.sizeof(opt), None)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        assert rc == 0, rc
        buf = ctypes.create_string_buffer(1400)
        _lib.lib.gespmm_plan_describe(h, buf, 1400)
        desc = buf.value.decode()[:150]
        _lib.lib.gespmm_plan_destroy(h)
    print("%-5s %-16s gespmm_init %.1f ms | plan creations %s ms | %s" % (mode, name, t_init, " ".join("%.2f" % t for t in ts), desc), flush=True)


if len(sys.argv) > 2 and sys.argv[1] == "--child":
    child(sys.argv[2], sys.argv[3])
    sys.exit(0)

names = sys.argv[1:] or ["com-amazon-sbm"]
for name in names:
    for mode in ("cold", "warm", "cold", "warm"):
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", mode, name], check=False)
# the reference's protocol through the driver: fresh process per run
import torch  # noqa: E402

from gespmm_amd import graphs  # noqa: E402

drv = os.path.join(ROOT, "gespmm_amd", "lib", "spmm_test")
for name in names:
    g = graphs.synthetic_graph(name, seed=42, device="cuda")
    path = "/tmp/%s.mtx" % name
    graphs.write_mtx(path, g["rowptr"], g["colind"])
    del g
    torch.cuda.empty_cache()
    for extra in ([], ["--method", "-1"], ["--plan"], ["--plan", "--method", "-1"]):
        r = subprocess.run([drv, path, "0", "--ncols", "128", "--no-vendor", "--seed", "1", "--out", "/tmp/plan_cold.out"] + extra,
                           capture_output=True, text=True)
        keep = [ln for ln in r.stdout.splitlines() if ln.startswith("N=128") or ln.startswith("gespmm_init")]
        print("driver %-24s | %s" % (" ".join(extra) or "(default: method 2)", " | ".join(k[:200] for k in keep)), flush=True)
        if r.returncode != 0:
            print(r.stderr[-400:])
