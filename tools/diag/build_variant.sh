#!/bin/bash
# Build a variant of the library with extra -D flags on fr_blend.hip (HERE, in the build container):
#   tools/diag/build_variant.sh <name> -DFR_DIAG_TRACE ...   ->  .ab/libfr_<name>.so     (fr_diag.hpp lists the switches)
# The other objects come from the regular build (run `make -C fateavatar_amd/csrc` first).
set -e
name=$1; shift
cd "$(dirname "$0")/../../fateavatar_amd/csrc"
mkdir -p ../../.ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -munsafe-fp-atomics \
    -ffp-contract=fast -fno-slp-vectorize "$@" -c fr_blend.hip -o /tmp/fr_blend_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../.ab/libfr_$name.so /tmp/fr_blend_$name.o \
    _obj/fr_preprocess.o _obj/fr_preprocess_bwd.o _obj/fr_knn.o _obj/fr_optim.o _obj/fr_binding.o _obj/fr_api.o
echo built .ab/libfr_$name.so
