"""The compiled `_C` module (torch C++ extension) over the C ABI: build and load.

`fateavatar_amd/csrc/torch_ext.cpp` exports what the reference's pybind module exports
(submodules/diff-gaussian-rasterization/ext.cpp:15-19: rasterize_gaussians, rasterize_gaussians_backward,
mark_visible; submodules/simple-knn/ext.cpp: distCUDA2) with the reference's C++ signatures.  It is host code only
(torch glue around libfr_hip.so), so the host compiler builds it; the shared object stays in-tree
(fateavatar_amd/_torch_ext/fr_torch_C.so) and travels with the repository snapshot.

The default Python host (fateavatar_amd/rasterizer.py) talks to the same library through ctypes and adds the
extensions the training step uses (gradient slots, fused activations / statistics, no-wait capture); this module is
the drop-in for callers that want the reference's `_C` surface itself: `FR_USE_TORCH_EXT=1` makes
`diff_gaussian_rasterization._C` / `simple_knn._C` resolve to it.
"""
from __future__ import annotations

import importlib.util
import os
import subprocess
import sys
import sysconfig

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "torch_ext.cpp")
OUT_DIR = os.path.join(_HERE, "_torch_ext")
NAME = "fr_torch_C"
SO_PATH = os.path.join(OUT_DIR, NAME + ".so")
_mod = None


def build(force: bool = False) -> str:
    """g++ -shared torch_ext.cpp against the installed PyTorch-ROCm headers and libfr_hip.so (about a minute)."""
    import torch
    from torch.utils import cpp_extension as ce
    from . import _lib
    # (the extension only links against libfr_hip.so's C ABI: a rebuilt library does not require a rebuilt extension)
    deps = [SRC, os.path.join(_HERE, "..", "include", "fr_rasterizer.h")]
    if not os.path.exists(_lib.SO_PATH):
        _lib.build()
    if not force and os.path.exists(SO_PATH) and all(os.path.getmtime(SO_PATH) >= os.path.getmtime(d) for d in deps):
        return SO_PATH
    os.makedirs(OUT_DIR, exist_ok=True)
    inc = [os.path.join(_HERE, "..", "include"), "/opt/rocm/include", sysconfig.get_paths()["include"]] + ce.include_paths()
    tl = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = (["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
            "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", f"-DTORCH_EXTENSION_NAME={NAME}", "-DTORCH_API_INCLUDE_EXTENSION_H",
            f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
           + [f"-I{p}" for p in inc] + [SRC, "-o", SO_PATH, f"-L{_HERE}", "-lfr_hip", "-Wl,-rpath,$ORIGIN/..", f"-L{tl}", "-lc10", "-lc10_hip",
                                        "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python"])
    subprocess.check_call(cmd)
    return SO_PATH


def load():
    """Import the built module.  libfr_hip.so is loaded first (RTLD_GLOBAL, by fateavatar_amd._lib), so the
    extension's dependency on it resolves without an rpath."""
    global _mod
    if _mod is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        import torch  # noqa: F401  (libtorch first)
        from . import _lib
        _lib.lib()
        spec = importlib.util.spec_from_file_location(NAME, SO_PATH)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        sys.modules.setdefault(NAME, mod)
        _mod = mod
    return _mod
