#!/bin/bash
# Build a variant of the WHOLE library with extra -D flags on every translation unit (switches that live in fr_common.hpp):
#   tools/diag/build_variant_all.sh <name> -DFR_PRE_WG=64 ...   ->  .ab/libfr_<name>.so
set -e
name=$1; shift
cd "$(dirname "$0")/../../fateavatar_amd/csrc"
mkdir -p ../../.ab /tmp/fr_$name
objs=""
for u in fr_preprocess fr_preprocess_bwd fr_knn fr_blend fr_optim fr_binding fr_api; do
  mode="-ffp-contract=off -fno-slp-vectorize"
  [ $u = fr_blend ] && mode="-ffp-contract=fast -fno-slp-vectorize"
  [ $u = fr_api ] && mode=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -munsafe-fp-atomics $mode "$@" -c $u.hip -o /tmp/fr_$name/$u.o &
  objs="$objs /tmp/fr_$name/$u.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../.ab/libfr_$name.so $objs
echo built .ab/libfr_$name.so
