#!/bin/bash
# Diagnostic builds of libmad_amd.so with other compiler flags, next to the product build (which stays untouched):
#   tools/build_variant.sh <name> <extra / replacement flags...>   ->  mad_amd/csrc/build_<name>/libmad_amd_<name>.so
# Use with MAD_LIB_PATH=<that file>.  The ISA of every kernel is kept beside it (-save-temps).
name=$1; shift
cd "$(dirname "$0")/../mad_amd/csrc" || exit 1
mkdir -p build_$name
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -save-temps=obj -Wno-unused-function"
for f in mad_ctx mad_orient mad_match mad_refine mad_space; do
  /opt/rocm/bin/hipcc $BASE "$@" -c $f.hip -o build_$name/$f.o 2> build_$name/$f.log || { tail -5 build_$name/$f.log; exit 1; }
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_$name/libmad_amd_$name.so build_$name/mad_ctx.o build_$name/mad_orient.o build_$name/mad_match.o build_$name/mad_refine.o build_$name/mad_space.o && ls -la build_$name/libmad_amd_$name.so
grep -c "v_pk_[a-z0-9]*_f32" build_$name/*gfx950.s
