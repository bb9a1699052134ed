#!/usr/bin/env python3
"""Ablation builds of the narrow staged kernel (where do its microseconds go at N = 32?): the product source is patched into
profiles/r05/experiments/_build/<case>/ — never in gespmm_amd/csrc — and linked with the product's other objects. The results of
these libraries are WRONG on purpose (a part of the work is left out); only their times are read.
    python profiles/r05/experiments/narrow_ablate_build.py        (needs gespmm_amd/lib/obj/*.o: run __graft_entry__.build() first)"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
CSRC = os.path.join(ROOT, "gespmm_amd", "csrc")
OUT = os.path.join(ROOT, "profiles", "r05", "experiments", "_build")
SRC = open(os.path.join(CSRC, "spmm_staged_narrow.hip")).read()

PATCH = {
    # entries whose B row is not staged read LDS like the others (garbage): no memory gathers at all
    "nomem": [("const uint64_t mm = __builtin_amdgcn_uicmp((uint32_t)code, (uint32_t)kStagedRowEnd, 36) & m_act;", "const uint64_t mm = 0;")],
    # row ends zero the accumulators but store nothing
    "nostore": [('                "global_store_dwordx4 %[t], v[60:63], %[C]\\n\\t"\n', "")],
    # the staging copy reads nothing from B (the LDS rows are zero)
    "nostage": [("if (hcol[u] >= 0) stage[u] = B4[(size_t)hcol[u] * W + (i % W)];", "")],
}
PATCH["all"] = PATCH["nomem"] + PATCH["nostore"] + PATCH["nostage"]
# the walk alone: no records are consumed (every group is done at once) — what a launch costs before its first window
PATCH["nowalk"] = [("const int gb = t.z, ge = t.w;", "const int gb = t.z, ge = t.z;")]

others = [o for o in glob.glob(os.path.join(ROOT, "gespmm_amd", "lib", "obj", "*.o")) if not o.endswith("spmm_staged_narrow.o")]
if len(others) < 10:
    sys.exit("build the product first (gespmm_amd/lib/obj is empty)")
for case, subs in PATCH.items():
    d = os.path.join(OUT, case)
    os.makedirs(d, exist_ok=True)
    s = SRC
    for a, b in subs:
        assert s.count(a) == 1, (case, a)
        s = s.replace(a, b)
    p = os.path.join(d, "spmm_staged_narrow.hip")
    open(p, "w").write(s)
    obj = os.path.join(d, "spmm_staged_narrow.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-I", CSRC, "-c", p, "-o", obj])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(d, "libgespmm.so"), obj] + others)
    print(case, "built")
