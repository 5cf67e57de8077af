#!/bin/bash
# k_describe's texel gathers with other cache-policy bits (probe builds made by
#   for a in 0 1 2 16 17 18 19; do tools/build_variant.sh aux$a -fno-slp-vectorize -fno-vectorize -DDSC_TEX_AUX=$a; done):
# per-kernel times of the serialised C3 bench for the product build and for every variant present.
cd "$GRAFT_REPO_ROOT" || exit 1
tools/kstats.sh aux_prod | grep "k_describe\|k_orient<"
for d in mad_amd/csrc/build_aux*; do
  a=${d##*build_aux}
  echo "== aux $a"
  tools/kstats.sh aux_$a MAD_LIB_PATH=$PWD/$d/libmad_amd_aux$a.so | grep "k_describe\|k_orient<" || exit 1
done
