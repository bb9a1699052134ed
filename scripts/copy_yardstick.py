#!/usr/bin/env python3
"""The streaming-copy yardstick (gespmm_baseline_copy_f32) in its launch shapes, and torch's own copy beside it: which one is the rate a
plain read + write reaches on this box. GESPMM_COPY_MODE is read once per process: the script re-executes itself per mode.
    python scripts/copy_yardstick.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1:
    import torch

    from gespmm_amd import spmm

    mode = sys.argv[1]
    for mb in (200, 800):
        n = mb * 250 * 1000
        src = torch.empty(n, dtype=torch.float32, device="cuda").uniform_(-1, 1)
        dst = torch.empty_like(src)
        fn = (lambda: dst.copy_(src)) if mode == "torch" else (lambda: spmm.baseline_copy(src, out=dst))
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print("mode %-5s %4d MB: %8.1f us  %.2f TB/s read + write  %s" % (mode, mb, us, 8.0 * n / us / 1e6, "ok" if torch.equal(src, dst) else "WRONG"),
              flush=True)
        del src, dst
else:
    for mode in ("0", "1", "2", "3", "4", "5", "torch"):
        env = dict(os.environ)
        if mode != "torch":
            env["GESPMM_COPY_MODE"] = mode
        subprocess.run([sys.executable, os.path.abspath(__file__), mode], env=env, check=False)
