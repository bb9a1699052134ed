#!/usr/bin/env python3
"""Experiment: cache-policy bits of the C row stores of the staged-rows kernel (product: sc1). Variants built into
profiles/r06/experiments/_build/store_<name>/libgespmm.so.     python profiles/r06/experiments/staged_store_scope_build.py"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
CSRC = os.path.join(ROOT, "gespmm_amd", "csrc")
src = open(os.path.join(CSRC, "spmm_staged.hip")).read()
others = [o for o in glob.glob(os.path.join(ROOT, "gespmm_amd", "lib", "obj", "*.o")) if not o.endswith("spmm_staged.o")]
if len(others) < 10:
    sys.exit("build the product first (gespmm_amd/lib/obj is empty)")
for name, mod in (("none", ""), ("nt", " nt"), ("sc1nt", " sc1 nt"), ("sc0sc1", " sc0 sc1"), ("sc0sc1nt", " sc0 sc1 nt")):
    s = src
    for old in ("%[C] sc1\\n\\t", '" BASE " sc1\\n\\t'):
        assert s.count(old) == 1, old
        s = s.replace(old, old.replace(" sc1", mod))
    out = os.path.join(ROOT, "profiles", "r06", "experiments", "_build", "store_" + name)
    os.makedirs(out, exist_ok=True)
    p = os.path.join(out, "spmm_staged.hip")
    open(p, "w").write(s)
    obj = os.path.join(out, "spmm_staged.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-I", CSRC, "-c", p, "-o", obj])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, "libgespmm.so"), obj] + others)
    print("store", name, "ok", flush=True)
