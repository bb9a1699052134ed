"""The PyTorch-ROCm extension module `_gespmm_torch` (csrc/torch_binding.cpp): build
recipe and loader.

`build()` compiles it in-tree with plain g++ against the torch headers (the file has no
device code; it only calls the C ABI of libgespmm.so) into gespmm_amd/lib/, next to
libgespmm.so. `ext` is the imported module, or None when it has not been built — the
callers then use the ctypes binding of the same C ABI (`_lib.py`), never a CPU path.
"""
import importlib.util
import os
import subprocess
import sys
import sysconfig

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "csrc", "torch_binding.cpp")
_LIBDIR = os.path.join(_HERE, "lib")
_NAME = "_gespmm_torch"
EXT_PATH = os.path.join(_LIBDIR, _NAME + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def _up_to_date():
    if not os.path.exists(EXT_PATH):
        return False
    t = os.path.getmtime(EXT_PATH)
    deps = [_SRC, os.path.join(_HERE, "..", "include", "gespmm.h")]
    return all(os.path.getmtime(d) <= t for d in deps if os.path.exists(d))


def build(verbose=False):
    """Compile the extension (about half a minute; skipped when up to date)."""
    if _up_to_date():
        return EXT_PATH
    import torch
    from torch.utils import cpp_extension as ce

    inc = []
    for p in ce.include_paths(device_type="cuda") + [sysconfig.get_paths()["include"]]:
        inc += ["-I", p]
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", _SRC, "-o", EXT_PATH,
           "-DTORCH_EXTENSION_NAME=" + _NAME, "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)] + inc + [
           "-L", _LIBDIR, "-lgespmm", "-L", torch_lib, "-lc10", "-ltorch", "-ltorch_cpu", "-ltorch_python",
           "-lc10_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + torch_lib]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return EXT_PATH


def _load():
    if not os.path.exists(EXT_PATH) or os.environ.get("GESPMM_NO_TORCH_EXT"):
        return None
    import torch  # noqa: F401  (libtorch must be loaded before the extension)

    spec = importlib.util.spec_from_file_location(_NAME, EXT_PATH)
    mod = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(mod)
    except ImportError as e:  # stale build against another torch: fall back to ctypes, loudly
        sys.stderr.write("gespmm_amd: could not load %s (%s); using the ctypes binding\n" % (EXT_PATH, e))
        return None
    return mod


ext = _load()
