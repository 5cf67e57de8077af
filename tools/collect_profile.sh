#!/bin/bash
# Copies what tools/profile_round.sh left in gpurun_out/<tag>/ into profiles/ under the round's name:
#   tools/collect_profile.sh <tag> <name>      e.g.  r02a_prof r02_a
tag=$1; name=$2
cd "$(dirname "$0")/.." || exit 1
cp gpurun_out/$tag/summary.json profiles/${name}_bench_c3_serial_summary.json
cp "$(find gpurun_out/$tag/trace -name '*kernel_stats.csv' | head -1)" profiles/${name}_bench_c3_serial_kernel_stats.csv
cp gpurun_out/$tag/line.json profiles/${name}_bench_c3_line.json
ls -la profiles/${name}_*
