"""`from simple_knn._C import distCUDA2` (reference submodules/simple-knn/ext.cpp)."""
from fateavatar_amd.knn import distCUDA2  # noqa: F401
