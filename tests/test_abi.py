"""CPU: the C-ABI library loads here (no GPU needed) and exports every symbol include/fr_rasterizer.h
declares; the ctypes mirrors of the ABI structs have the C compiler's layout."""
import ctypes as C
import os
import re
import subprocess
import tempfile

from fateavatar_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "fr_rasterizer.h")


def _declared_functions():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fr_[A-Za-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    names = _declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/fr_rasterizer.h but not exported"
    assert set(_lib.EXPORTS) == set(names)


def test_struct_layouts_match_the_c_compiler():
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "fr_rasterizer.h"
int main(void){
 printf("%zu %zu %zu %zu\n", sizeof(fr_params), sizeof(fr_inputs), sizeof(fr_grads), sizeof(fr_counts));
 printf("%zu %zu %zu\n", offsetof(fr_params, tan_fovx), offsetof(fr_params, debug), offsetof(fr_inputs, campos));
 printf("%zu %zu %zu %zu\n", sizeof(fr_aux), sizeof(fr_binding), offsetof(fr_aux, binding), offsetof(fr_aux, d_scaling));
 printf("%zu %zu %zu %zu\n", offsetof(fr_aux, overflow_out), sizeof(fr_adam_config), offsetof(fr_adam_config, skip), offsetof(fr_adam_config, n_skip));
 return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write(prog)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        out = subprocess.check_output([exe]).decode().split()
    sizes = [int(x) for x in out]
    assert sizes[:4] == [C.sizeof(_lib.fr_params), C.sizeof(_lib.fr_inputs), C.sizeof(_lib.fr_grads), C.sizeof(_lib.fr_counts)]
    assert sizes[4] == _lib.fr_params.tan_fovx.offset
    assert sizes[5] == _lib.fr_params.debug.offset
    assert sizes[6] == _lib.fr_inputs.campos.offset
    assert sizes[7:11] == [C.sizeof(_lib.fr_aux), C.sizeof(_lib.fr_binding), _lib.fr_aux.binding.offset, _lib.fr_aux.d_scaling.offset]
    assert sizes[11:15] == [_lib.fr_aux.overflow_out.offset, C.sizeof(_lib.fr_adam_config), _lib.fr_adam_config.skip.offset,
                            _lib.fr_adam_config.n_skip.offset]


def test_scratch_size_queries_are_sane_without_a_gpu():
    L = _lib.lib()
    assert L.fr_version().startswith(b"fateavatar_amd")
    g1, g2 = L.fr_geometry_bytes(1000), L.fr_geometry_bytes(2000)
    assert 0 < g1 < g2 <= 2 * g1 + 4096
    assert L.fr_image_bytes(512, 512) > 512 * 512 * 8
    assert L.fr_binning_bytes(1000, 64, 64) >= 1000 * 56
    assert L.fr_knn_workspace_bytes(100000) > 100000 * 16


def test_header_cites_the_reference_interfaces():
    src = open(HDR).read()
    for needle in ["rasterizer_impl.cu:198-336", "rasterizer_impl.cu:340-434", "rasterize_points.cu:198-217",
                   "simple_knn.cu:186-222", "spatial.cu:14-25"]:
        assert needle in src
