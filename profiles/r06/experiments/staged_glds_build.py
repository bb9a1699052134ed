#!/usr/bin/env python3
"""Experiment (round-5 review item 4): the staged-rows kernel stages its block's B rows with direct-to-LDS loads
(global_load_lds_dwordx4: no staging registers, no ds_write pass) from the SAME two-blocks-per-CU shape. The product source is patched
into profiles/r06/experiments/_build/glds/ — never in gespmm_amd/csrc.      python profiles/r06/experiments/staged_glds_build.py"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
CSRC = os.path.join(ROOT, "gespmm_amd", "csrc")
OUT = os.path.join(ROOT, "profiles", "r06", "experiments", "_build", "glds")
s = open(os.path.join(CSRC, "spmm_staged.hip")).read()


def rep(old, new):
    global s
    assert s.count(old) == 1, old
    s = s.replace(old, new)


# the loads: thread t's piece u goes to LDS piece u * T + t = (wave-uniform base u * T + wave * 64) + lane: lane-linear, as the DMA writes
rep("""    f4v stage[P];
#pragma unroll
    for (int u = 0; u < P; ++u) {
        const int i = u * kStagedWaves * 64 + tid;
        stage[u] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
        if (hcol[u] >= 0) stage[u] = B4[(((size_t)hcol[u] << TSHIFT) + (size_t)tile) * kRowF4 + (i % kRowF4)];
    }
""", """#pragma unroll
    for (int u = 0; u < P; ++u) {
        const int i = u * kStagedWaves * 64 + tid;
        if (hcol[u] >= 0)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(B4 + ((((size_t)hcol[u] << TSHIFT) + (size_t)tile) * kRowF4 + (i % kRowF4))),
                (__attribute__((address_space(3))) void*)(s_hot + (u * kStagedWaves * 64 + wave * 64)), 16, 0, 0);
    }
""")
rep("""#pragma unroll
    for (int u = 0; u < P; ++u)
        if (hcol[u] >= 0) s_hot[u * kStagedWaves * 64 + tid] = stage[u];
    __syncthreads();
""", """    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA writes are counted with the vector loads
    __syncthreads();
""")
os.makedirs(OUT, exist_ok=True)
p = os.path.join(OUT, "spmm_staged.hip")
open(p, "w").write(s)
others = [o for o in glob.glob(os.path.join(ROOT, "gespmm_amd", "lib", "obj", "*.o")) if not o.endswith("spmm_staged.o")]
if len(others) < 10:
    sys.exit("build the product first (gespmm_amd/lib/obj is empty)")
obj = os.path.join(OUT, "spmm_staged.o")
flags = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-I", CSRC]
subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", p, "-o", obj])
subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-S", "--cuda-device-only", p, "-o", os.path.join(OUT, "spmm_staged.s")])
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, "libgespmm.so"), obj] + others)
print("glds build ok")
