#!/bin/bash
# Build a variant of the library with extra -D flags on ONE translation unit (HERE, in the build container):
#   tools/diag/build_variant_file.sh <name> <fr_preprocess|fr_preprocess_bwd|fr_blend|fr_api> -DFR_DIAG_... ->  .ab/libfr_<name>.so
# The other objects come from the regular build (run `make -C fateavatar_amd/csrc` first).  fr_diag.hpp lists the switches.
set -e
name=$1; unit=$2; shift; shift
cd "$(dirname "$0")/../../fateavatar_amd/csrc"
mkdir -p ../../.ab
mode="-ffp-contract=off -fno-slp-vectorize"
[ $unit = fr_blend ] && mode="-ffp-contract=fast -fno-slp-vectorize"
[ $unit = fr_api ] && mode=""
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -munsafe-fp-atomics $mode "$@" -c $unit.hip -o /tmp/${unit}_$name.o
objs=""
for u in fr_preprocess fr_preprocess_bwd fr_knn fr_blend fr_optim fr_binding fr_api; do
  if [ $u = $unit ]; then objs="$objs /tmp/${unit}_$name.o"; else objs="$objs _obj/$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../.ab/libfr_$name.so $objs
echo built .ab/libfr_$name.so
