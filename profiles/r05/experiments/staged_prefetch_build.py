#!/usr/bin/env python3
"""Experiment: a block of the staged-rows kernel touches the staging list and the tasks of the block D positions ahead of it (same XCD
slice), so that block's first round trip finds them in L2 / the Infinity Cache instead of in HBM. The product source is patched into
profiles/r05/experiments/_build/prefetch/ — never in gespmm_amd/csrc. D comes from GESPMM_STAGED_PF at launch (0 = off).
    python profiles/r05/experiments/staged_prefetch_build.py"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
CSRC = os.path.join(ROOT, "gespmm_amd", "csrc")
OUT = os.path.join(ROOT, "profiles", "r05", "experiments", "_build", "prefetch")
s = open(os.path.join(CSRC, "spmm_staged.hip")).read()


def rep(old, new):
    global s
    assert s.count(old) == 1, old
    s = s.replace(old, new)


rep("    a.debug = dbg_env;\n",
    '    static const int pf_env = getenv("GESPMM_STAGED_PF") ? atoi(getenv("GESPMM_STAGED_PF")) : 0;\n    a.debug = pf_env << 8;\n')
# the touch: issued right behind the block's own first round trip ...
rep("    const int wb = tk[2], we = tk[3];  // the wavefront's range",
    "    int pf_v = 0;\n"
    "    {\n"
    "        const int pb = blk + (a.debug >> 8);\n"
    "        if ((a.debug >> 8) > 0 && pb < a.nblocks) {\n"
    "            if (wave == 0 && lane * 16 < H * 4) pf_v = *reinterpret_cast<const volatile int*>(reinterpret_cast<const char*>(a.hot_cols + (size_t)pb * H) + lane * 16);\n"
    "            if (wave == 1 && lane * 64 < kStagedWaves * 16) pf_v = *reinterpret_cast<const volatile int*>(reinterpret_cast<const char*>(a.tasks) + ((size_t)pb * kStagedWaves * 16 + lane * 64));\n"
    "        }\n"
    "    }\n"
    "    const int wb = tk[2], we = tk[3];  // the wavefront's range")
# ... and waited for where everything older has been waited for anyway
rep("    __syncthreads();\n    __builtin_amdgcn_s_waitcnt(0);  // the compiler's scoreboard is clean when the assembly gathers start\n",
    "    __syncthreads();\n    __builtin_amdgcn_s_waitcnt(0);  // the compiler's scoreboard is clean when the assembly gathers start\n    asm volatile(\"\" ::\"v\"(pf_v));\n")
os.makedirs(OUT, exist_ok=True)
p = os.path.join(OUT, "spmm_staged.hip")
open(p, "w").write(s)
others = [o for o in glob.glob(os.path.join(ROOT, "gespmm_amd", "lib", "obj", "*.o")) if not o.endswith("spmm_staged.o")]
if len(others) < 10:
    sys.exit("build the product first (gespmm_amd/lib/obj is empty)")
obj = os.path.join(OUT, "spmm_staged.o")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-I", CSRC, "-c", p, "-o", obj])
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, "libgespmm.so"), obj] + others)
print("prefetch build ok")
