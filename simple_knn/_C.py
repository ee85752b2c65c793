"""`from simple_knn._C import distCUDA2` (reference submodules/simple-knn/ext.cpp).  FR_USE_TORCH_EXT=1 selects the
compiled torch extension (fateavatar_amd/csrc/torch_ext.cpp) instead of the ctypes host."""
import os

if os.environ.get("FR_USE_TORCH_EXT") == "1":
    from fateavatar_amd import torch_ext as _te
    distCUDA2 = _te.load().distCUDA2
else:
    from fateavatar_amd.knn import distCUDA2  # noqa: F401
