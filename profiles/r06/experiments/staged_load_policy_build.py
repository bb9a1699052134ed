#!/usr/bin/env python3
"""Experiment (after `sc1 nt` C stores went in): cache-policy bits of the staged-rows kernel's LOADS — the memory gathers of entries that
are not staged (G), the staging copy of the block's rows (S). Variants in profiles/r06/experiments/_build/load_<name>/libgespmm.so."""
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


def gathers(s, mod):
    for old in ('"global_load_dwordx4 %0, %2, %1\\n\\t"', '"global_load_dwordx2 %0, %1, %3\\n\\t"', '"global_load_dwordx4 %0, %1, %3\\n\\t"'):
        assert s.count(old) == 1, old
        s = s.replace(old, old.replace("\\n\\t", mod + "\\n\\t"))
    return s


def staging(s):
    old = "if (hcol[u] >= 0) stage[u] = B4[(((size_t)hcol[u] << TSHIFT) + (size_t)tile) * kRowF4 + (i % kRowF4)];"
    assert s.count(old) == 1
    return s.replace(old, "if (hcol[u] >= 0) stage[u] = __builtin_nontemporal_load(&B4[(((size_t)hcol[u] << TSHIFT) + (size_t)tile) * kRowF4 + (i % kRowF4)]);")


for name, fn in (("Gnt", lambda s: gathers(s, " nt")), ("Gsc1", lambda s: gathers(s, " sc1")), ("Snt", staging), ("GntSnt", lambda s: staging(gathers(s, " nt")))):
    out = os.path.join(ROOT, "profiles", "r06", "experiments", "_build", "load_" + name)
    os.makedirs(out, exist_ok=True)
    p = os.path.join(out, "spmm_staged.hip")
    open(p, "w").write(fn(src))
    obj = os.path.join(out, "spmm_staged.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-I", CSRC, "-c", p, "-o", obj])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, "libgespmm.so"), obj] + others)
    print("load", name, "ok", flush=True)
